"""D2M distillation losses (reference: distillers.py).  Same `Distiller(name, cfg, device)` object and
method names; each method is resolved by `getattr(distiller, config.distill_name)` (trainwandb.py:231)
and returns a dict with at least 'loss' (0-dim tensor carrying grad).

All methods built from the three primitives kd_loss / inter_class_relation / cross_entropy run as ONE
fused HIP launch (values + logits gradients).  Methods of the reference that need other plugins
(strm*, KL_feature, wsl focal weighting) raise NotImplementedError."""
from . import ops


def _terms(s_kl=None, t_kl=None, s_ce=None, labels=None, s_sup=None, t_sup=None, T=4.0, w_kl=0.0, w_sup=0.0, w_ce=0.0):
    out = ops.D2MLossFn.apply(s_kl, t_kl, s_ce, labels, s_sup, t_sup, float(T), float(w_kl), float(w_sup), float(w_ce))
    return out[0], out[1].detach(), out[2].detach(), out[3].detach()


def kd_loss(logits_student, logits_teacher, temperature):
    """distillers.py:7-15"""
    return _terms(s_kl=logits_student, t_kl=logits_teacher, T=temperature, w_kl=1.0)[0]


def inter_class_relation(y_s, y_t):
    """distillers.py:18-30 (softmax -> Pearson correlation -> 1 - mean)"""
    return _terms(s_sup=y_s, t_sup=y_t, w_sup=1.0)[0]


def cross_entropy(logits, labels):
    return _terms(s_ce=logits, labels=labels, w_ce=1.0)[0]


class Distiller(object):
    def __init__(self, distill_name, distill_cfg, device):
        self.distill_name = distill_name
        self.distill_dict = distill_cfg
        self.device = device

    def _to(self, t):
        return t.to(self.device)

    def KD(self, student_logits, teacher_logits, test_labels):
        """distillers.py:42-74"""
        d = self.distill_dict
        s, t = self._to(student_logits), self._to(teacher_logits)
        w_ce, w_kl = d["hard_loss_weight"] / 16.0, d["soft_loss_weight"]
        loss, kl, _, ce = _terms(s_kl=s, t_kl=t, s_ce=s, labels=test_labels, T=d["temperature"], w_kl=w_kl, w_ce=w_ce)
        return {"hard_loss": w_ce * ce, "soft_loss": w_kl * kl, "loss": loss}

    def ce(self, student_logits, teacher_logits, test_labels):
        """distillers.py:100-108"""
        s = self._to(student_logits)
        loss = _terms(s_ce=s, labels=test_labels, w_ce=self.distill_dict["hard_loss_weight"] / 16.0)[0]
        return {"loss": loss}

    def fc_2(self, student_logits, teacher_logits, test_labels):
        """distillers.py:152-161"""
        d = self.distill_dict
        t = self._to(teacher_logits)
        fc1, fc2 = self._to(student_logits["fc_1"]), self._to(student_logits["fc_2"])
        w_ce, w_kl = d["hard_loss_weight"] / 16.0, d["soft_loss_weight"]
        loss, kl, _, ce = _terms(s_kl=fc2, t_kl=t, s_ce=fc1, labels=test_labels, T=d["temperature"], w_kl=w_kl, w_ce=w_ce)
        return {"hard_loss": w_ce * ce, "soft_loss": w_kl * kl, "loss": loss}

    def Dist_KD(self, student_logits, teacher_logits, test_labels):
        """distillers.py:286-293"""
        d = self.distill_dict
        s, t = self._to(student_logits), self._to(teacher_logits)
        w_ce, w_sup = d["hard_loss_weight"] / 16.0, d["soft_loss_weight"]
        loss, _, sup, ce = _terms(s_ce=s, labels=test_labels, s_sup=s, t_sup=t, w_sup=w_sup, w_ce=w_ce)
        return {"soft_loss": w_sup * sup, "hard_loss": w_ce * ce, "loss": loss}

    def fc_2_sup_dist(self, student_logits, teacher_logits, test_labels):
        """distillers.py:295-337 — the default: kd_loss(kl) + 0.5*inter_class_relation(sup) + CE(ce)/16"""
        s_kl, s_sup, s_ce = (self._to(student_logits[k]) for k in ("kl", "sup", "ce"))
        t_kl, t_sup = self._to(teacher_logits["kl"]), self._to(teacher_logits["sup"])
        loss, kl, sup, ce = _terms(s_kl=s_kl, t_kl=t_kl, s_ce=s_ce, labels=test_labels, s_sup=s_sup, t_sup=t_sup,
                                   T=self.distill_dict["temperature"], w_kl=1.0, w_sup=0.5, w_ce=1.0 / 16.0)
        return {"soft_loss": kl, "hard_loss": 0.5 * sup + ce / 16.0, "loss": loss}

    def e_dist_1fc_sup(self, student_logits, teacher_logits, test_labels):
        """distillers.py:713-733"""
        s_kl, s_sup = self._to(student_logits["kl"]), self._to(student_logits["sup"])
        t_kl, t_sup = self._to(teacher_logits["kl"]), self._to(teacher_logits["sup"])
        loss = _terms(s_kl=s_kl, t_kl=t_kl, s_ce=s_kl, labels=test_labels, s_sup=s_sup, t_sup=t_sup,
                      T=self.distill_dict["temperature"], w_kl=1.0, w_sup=0.5, w_ce=1.0 / 16.0)[0]
        return {"loss": loss}

    def __getattr__(self, name):
        if name in _OUT_OF_SCOPE:
            def _missing(*a, **k):
                raise NotImplementedError("Distiller.%s is outside the MI355X hot path (SURVEY.md 8f N2)" % name)
            return _missing
        raise AttributeError(name)


_OUT_OF_SCOPE = {"wsl", "support_sim", "KL_feature", "fc_2_wsl", "strm", "strm_KD", "fc_2_sup", "fc_2_sup_kl",
                 "fc_2_sup_dist_cece", "fc_2_sup_klklcece", "fc_2_sup_distdistcece", "fc_2_sup_2", "fc_2_sup_disver",
                 "fc_2_sup_dist_wsl", "strm_fc_2_sup_dist", "strm_1fc_sup", "fc_1_sup", "fc_sup"}
