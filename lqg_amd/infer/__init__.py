from lqg_amd.infer.mle import candidate_search, max_likelihood
from lqg_amd.infer.models import get_model_params, log_likelihood_objective
from lqg_amd.infer.utils import infer, sample_from_prior
from lqg_amd.infer.gradient import value_and_grad
