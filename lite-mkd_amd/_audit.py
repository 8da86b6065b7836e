"""Stream-lifetime audit (LMKD_STREAM_AUDIT=1; tests turn it on for whole episodes).

torch's caching allocator hands a freed block back to the stream it was allocated on at once; a kernel that another stream has queued
on that block but not yet run then reads recycled memory.  The protection is `tensor.record_stream(other)` (or keeping the tensor alive
until an event of the other stream has passed) - per tensor, by hand, and easy to forget for the small ones: round 3 lost a BatchNorm
table that way, round 4 the words of a tensor's maximum (NaN weights after the first optimizer step with every parity test green).

With the audit on, every tensor this package allocates (torch.empty / empty_like / zeros / zeros_like / ones: wrapped) is remembered by its
storage address with the stream it was allocated on - from arm() on: what exists before (parameters, buffers, the flat buffers) lives as long as the run -;
`record_stream` (wrapped) adds streams to it; and every entry-point call made through
`_lib.call` checks the tensors whose addresses it was given (ops._p / ops._pw note them): a tensor used on a stream other than its own
without a record for that stream is a FINDING (entry point, argument index, shape, allocation stream, launch stream).  Tensors that were
not allocated through the wrapped factories - parameters, user inputs, autograd's gradients - are caller-owned and not judged.  A use that is
safe for another reason (the tensor is kept alive until the consumer stream has been joined) is declared with `audit_ok(tensor, why)`.
`findings()` returns the list; STRICT raises at the call."""
import os

import torch

ON = os.environ.get("LMKD_STREAM_AUDIT", "0") == "1"
ARMED = False      # allocations are remembered only while armed: what exists before arm() - parameters, buffers, the flat gradient bucket -
                   # lives as long as the run and is caller-owned
STRICT = os.environ.get("LMKD_STREAM_AUDIT_STRICT", "0") == "1"
_table = {}        # storage address -> [allocation stream, {recorded streams}, declared-safe reason | None, storage bytes]
_open = {}         # (storage address, stream) -> findings waiting for a record_stream (the package records AFTER it launches, which is
                   # as good as before: what matters is that the record exists when the tensor is freed)
_pending = []      # tensors named to the upcoming entry-point call
_findings = []
_installed = [False]


def _cur():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _key(t):
    return t.untyped_storage().data_ptr()


def arm(on=True):
    """start (stop) remembering allocations: call after the model, the optimizer and their flat buffers exist"""
    global ARMED
    ARMED = bool(on)


def _tag(t):
    if ARMED and torch.is_tensor(t) and t.is_cuda and t.numel() > 0:
        st = t.untyped_storage()
        k = st.data_ptr()
        for ok in [o for o in _open if o[0] == k]:      # the address is handed out again: the previous tensor died without the record
            _findings.extend(_open.pop(ok))
        _table[k] = [_cur(), set(), None, st.nbytes()]
    return t


def install():
    """wrap the allocation factories and Tensor.record_stream (idempotent)"""
    if _installed[0]:
        return
    _installed[0] = True
    for name in ("empty", "empty_like", "zeros", "zeros_like", "ones", "full"):
        orig = getattr(torch, name)

        def wrapped(*a, __orig=orig, **k):
            return _tag(__orig(*a, **k))
        setattr(torch, name, wrapped)
    orig_rec = torch.Tensor.record_stream

    def record_stream(self, s):
        if self.is_cuda and self.numel() > 0:
            k = _key(self)
            e = _table.get(k)
            if e is not None:
                e[1].add(s.cuda_stream)
            _open.pop((k, s.cuda_stream), None)      # recorded after the launch: resolved
        return orig_rec(self, s)
    torch.Tensor.record_stream = record_stream


def enable(on=True, strict=False):
    global ON, STRICT
    ON, STRICT = bool(on), bool(strict)
    if ON:
        install()
    del _pending[:]


def audit_ok(t, why):
    """this tensor's uses on other streams are safe for the stated reason (kept alive until those streams are joined)"""
    if ON and torch.is_tensor(t) and t.is_cuda:
        e = _table.get(_key(t))
        if e is not None:
            e[2] = why


def engine_owned(*tensors):
    """gradients the autograd engine hands to a backward function: where it passes one between streams it records the consumer stream on
    it itself (torch/csrc/autograd/input_buffer.cpp), out of this module's sight"""
    if ON:
        for t in tensors:
            audit_ok(t, "autograd engine: records the consumer stream on the gradients it passes between streams")


def note(t):
    if ON and t is not None:
        _pending.append(t)


def check(name):
    if not ON:
        return
    if not _pending:
        return
    cur = _cur()
    for i, t in enumerate(_pending):
        if not (torch.is_tensor(t) and t.is_cuda) or t.numel() == 0:
            continue
        st = t.untyped_storage()
        e = _table.get(st.data_ptr())
        if e is None or e[3] != st.nbytes() or e[0] == cur or cur in e[1] or e[2] is not None:      # (another size: the address belongs to an untracked tensor now)
            continue
        f = (name, i, tuple(t.shape), str(t.dtype), e[0], cur)
        _open.setdefault((st.data_ptr(), cur), []).append(f)
        if STRICT:
            del _pending[:]
            raise RuntimeError("stream-lifetime audit: %s argument %d %s %s allocated on stream %#x is used on stream %#x without "
                               "record_stream" % f)
    del _pending[:]


def findings(clear=False):
    """the uses that never got their record_stream: closed ones (the address was handed out again) and the ones still open"""
    out = list(_findings) + [f for fs in _open.values() for f in fs]
    if clear:
        del _findings[:]
        _open.clear()
    return out


def summary():
    """findings grouped by (entry point, argument index): [(count, entry point, argument, example shape)]"""
    g = {}
    for name, i, shape, dt, a, c in findings():
        k = (name, i)
        g.setdefault(k, [0, shape])[0] += 1
    return sorted(((v[0], k[0], k[1], v[1]) for k, v in g.items()), reverse=True)


if ON:
    install()
