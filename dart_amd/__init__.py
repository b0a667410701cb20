import os as _os
# one hardware queue per stream of the contexts in flight (HIP default: 4 queues shared by all streams); effective only
# if set before the HIP runtime initialises, hence here, at package import
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
