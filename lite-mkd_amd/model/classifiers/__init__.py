"""Classifier plugins (reference: model/classifiers/__init__.py).  Names resolved by
model_select.select_model_student / select_model_teacher via getattr on this package."""
from .TRX_2fcsup import (PositionalEncoding, SupportDK, TemporalCrossTransformer, TRX, TRX_2fc, TRX_2fcsup,  # noqa: F401
                         TRX_2fcsup_fixed, TRX_fixed, TRX_sup, TRX_sup_fixed)
from .e_dist_fc2 import CosDistance, e_dist, e_dist_1fc_sup, e_dist_fc2, e_dist_fc2_sup, e_dist_fc2_sup_fixed  # noqa: F401

__all__ = ["TRX", "TRX_fixed", "TRX_sup", "TRX_sup_fixed", "TRX_2fc", "TRX_2fcsup", "TRX_2fcsup_fixed", "e_dist", "e_dist_fc2", "e_dist_fc2_sup", "e_dist_fc2_sup_fixed",
           "e_dist_1fc_sup", "SupportDK", "TemporalCrossTransformer", "PositionalEncoding"]
