from lqg_amd.infer.mle import candidate_search, max_likelihood
from lqg_amd.infer.models import (common_objective, get_model_params, lifted_model, log_likelihood_objective, lqg_model,
                                  shared_params_objective, split_params)
from lqg_amd.infer.utils import infer, sample_from_prior
from lqg_amd.infer.gradient import value_and_grad
