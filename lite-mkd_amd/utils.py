"""utils.py of the reference, hot-path subset."""
from . import ops


def aggregate_accuracy(test_logits_sample, test_labels):
    """utils.py:116-121 — mean(argmax(logits,-1) == labels), computed by the HIP accuracy kernel."""
    acc, _ = ops.accuracy(test_logits_sample, None, test_labels)
    return acc


def aggregate_accuracy2(logits_a, logits_b, test_labels):
    """accuracy of logits_a + logits_b (trainwandb.py:247-257,278) without materialising the sum."""
    return ops.accuracy(logits_a, logits_b, test_labels)
