"""Stand-alone evaluator (reference: test.py:20-318 `Evaluator`, model/model_select.py:244-260 `select_test`): load a
checkpoint, run `num_test_tasks` test episodes in eval mode under no_grad (inference path: BatchNorm + residual + ReLU in the
convolution epilogue), print the per-task accuracy and return mean accuracy x100 with its 95 % confidence interval
196 * std / sqrt(n) (test.py:291-299).

    python -m litemkd_amd.evaluate --test_model student --test_model_path model_save/<ckpt>.pt --num_test_tasks 100

Data: any iterable of task dicts with the VideoDataset contract (video_reader.py:474-485); without one, synthetic test episodes
(trainloop.SyntheticEpisodes(train=False): 1 query per class, as `query_per_class_test`)."""
import numpy as np
import torch

from . import trainloop as TL
from .model import classifiers
from .model.model_select import load_student, load_teacher
from .options import default_args
from .utils import aggregate_accuracy


def select_test(args):
    """model_select.py:244-260.  'teacher': TRX_fixed with the `bracnch.transformers.0.*` weights of an MFM checkpoint
    (load_teacher); 'student': the checkpoint written by the training loop (load_student; commented out in the reference's
    dict although test.py:103-104 evaluates it)."""
    name2test = {
        "teacher": lambda: load_teacher(classifiers.TRX_fixed(args), args),
        "student": lambda: load_student(args),
    }
    return name2test[args.test_model]()


class Evaluator:
    def __init__(self, args, video_loader=None, log=print):
        self.args = args
        self.device = args.device
        self.log = log
        self.model = select_test(args).to(self.device)
        self.model.eval()
        self.video_loader = video_loader if video_loader is not None else TL.SyntheticEpisodes(args, base_seed=777, device=self.device, train=False)
        self.accuracy_fn = aggregate_accuracy

    def test(self):
        """test.py:65-299"""
        self.model.eval()
        accuracies = []
        with torch.no_grad():
            if hasattr(self.video_loader, "dataset"):
                self.video_loader.dataset.train = False
            iteration = 0
            for task_dict in self.video_loader:
                if iteration >= self.args.num_test_tasks:
                    break
                iteration += 1
                (context_images, target_images, context_teacher_feature, target_teacher_feature, context_labels, target_labels,
                 _, _) = TL.prepare_task(task_dict, self.device)
                if self.args.test_model == "student":
                    model_dict = self.model(context_images, context_labels, target_images)
                else:
                    model_dict = self.model(context_teacher_feature, context_labels, target_teacher_feature)
                logits = model_dict["logits"]
                if isinstance(logits, dict):         # two-head students: kl + ce as in the training loop (trainwandb.py:247-257)
                    logits = logits["kl"] + logits["ce"] if "ce" in logits else logits["kl"]
                accuracy = self.accuracy_fn(logits.to(self.device), target_labels)
                self.log("For Task: %s,  Testing Accuracy is %s" % (self.args.mode, accuracy.item()))
                accuracies.append(accuracy.item())
        accuracy = np.array(accuracies).mean() * 100.0
        confidence = (196.0 * np.array(accuracies).std()) / np.sqrt(len(accuracies))
        self.log("For Task: %s,  and Testing Accuracy is %s +- %s" % (self.args.mode, accuracy, confidence))
        return {self.args.dataset: {"accuracy": accuracy, "confidence": confidence}}


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    base = default_args()
    for k in ("test_model", "test_model_path", "teacher_checkpoint", "model_backbone", "model_classifier", "dataset", "mode"):
        ap.add_argument("--" + k, default=getattr(base, k))
    for k in ("num_test_tasks", "way", "shot", "query_per_class_test", "seq_len", "img_size"):
        ap.add_argument("--" + k, type=int, default=getattr(base, k))
    a = ap.parse_args(argv)
    args = default_args(**vars(a))
    print(Evaluator(args).test())


if __name__ == "__main__":
    main()
