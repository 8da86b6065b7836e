"""lite-mkd_amd — MI355X (gfx950) implementation of Lite-MKD's per-episode hot path.

Layout mirrors the reference for the path it replaces:
  model/model_select.py   Student / Teacher / select_model_student / select_model_teacher
  model/backbone/         resnet18_2fc, resnet18_student
  model/classifiers/      TRX_2fcsup(+_fixed), e_dist family, SupportDK
  distillers.py           Distiller (D2M losses)
  utils.py                aggregate_accuracy
  trainloop.py            the trainwandb.py episode loop (train / train_task / test / prepare_task)
  parallel.py             episode-parallel data parallelism: flat gradient bucket + RCCL all-reduce
  csrc/ + ops.py + _lib.py   HIP kernels, their C ABI (include/lmkd.h) and the autograd glue

Import name: `litemkd_amd` (the directory name has a hyphen; litemkd_amd.py at the repo root
loads this package)."""
from ._lib import lib, LIB_PATH  # noqa: F401

__all__ = ["lib", "LIB_PATH"]
