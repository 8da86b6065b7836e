"""Flag names and defaults of the reference's options.py:7-76 (hot-path subset), as a plain namespace."""
import argparse

import torch

DEFAULT_CFG = {"soft_loss_weight_support": 1, "soft_loss_weight_query": 1, "hard_loss_weight": 1, "soft_loss_weight": 2,
               "feature_loss_weight": 1, "temperature": 4, "fcwsl_aerfa": 0.5, "fcwsl_beta": 1}      # options.py:51-60


def default_args(**over):
    a = argparse.Namespace(
        way=5, shot=5, query_per_class=5, query_per_class_test=1, tasks_per_batch=16, print_freq=10, seq_len=8,
        num_workers=1, trans_linear_out_dim=1152, trans_linear_in_dim=2048, img_size=224, temp_set=[2], trans_dropout=0.1,
        save_freq=10000, split=3, sch=[20000, 40000], num_test_tasks=5000,
        device=torch.device("cuda" if torch.cuda.is_available() else "cpu"),
        method="resnet18", num_gpus=1, dataset="hmdb", mode="KD_KL_meta", debug=False,
        distill_name="fc_2_sup_dist", model_backbone="resnet18_2fc", model_classifier="TRX_2fcsup",
        model_teacher="test_teacher_TRX_2fcsup_fixed", teacher_checkpoint=None, test_model="student",
        soft_loss_weight=1, hard_loss_weight=1, test=False, cfg=dict(DEFAULT_CFG),
        checkpoint_dir=None, training_iterations=100010, learning_rate=0.0001, opt="sgd",
        test_iters=[], save_dir="model_save", test_model_path=None)
    for k, v in over.items():
        setattr(a, k, v)
    return a
