from lqg_amd.belief import kf  # noqa: F401
