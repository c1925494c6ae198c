from .detgen import det_uniform, det_normal, det_randint, det_param  # noqa: F401
