from lqg_amd.infer.mle import candidate_search, max_likelihood
from lqg_amd.infer.models import (common_objective, get_model_params, log_likelihood_objective,
                                  shared_params_objective, split_params)
from lqg_amd.infer.utils import infer, sample_from_prior
from lqg_amd.infer.gradient import value_and_grad
