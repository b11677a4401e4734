from lqg_amd.control import lqr  # noqa: F401
