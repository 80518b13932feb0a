"""Tensor-level wrappers and autograd Functions over the C ABI (include/pevae_hip.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only: every arithmetic op below
is a hand-written HIP kernel from libpevae_hip.so, launched on torch's current stream.  Nothing in this file
has a CPU path -- CPU tensors raise.
"""
import ctypes as C

import os

import torch

from . import _lib

BF16 = torch.bfloat16


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pevae_amd kernels need tensors on an MI355X device (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


def round_up(a, b):
    return (a + b - 1) // b * b


_WS = {}

TIMERS = None       # bench.py: dict tag -> [(start_event, end_event)] recorded on the launch stream


class timed(object):
    """HIP-event bracket around a kernel launch (active only while bench.py sets TIMERS)."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if TIMERS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if TIMERS is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            TIMERS.setdefault(self.tag, []).append((self.e0, e1))
        return False


_SIDE_STREAMS = {}


def parallel(*fns):
    """Run independent sub-graphs (callables) on separate HIP streams, forked from and joined back into the current
    stream.  Captured into a HIP graph they become parallel branches, and autograd replays each branch's backward on
    the stream its forward ran on: chains of tiny dependent kernels (the two DLS pyramids, the per-level Blocks of
    DLS_Auto) then overlap instead of leaving most CUs idle.  Outputs must stay referenced by the caller until they
    have been consumed (they are: autograd saves them)."""
    if len(fns) == 1:
        return [fns[0]()]
    cur = torch.cuda.current_stream()
    dev = cur.device
    pool = _SIDE_STREAMS.setdefault(dev, [])
    while len(pool) < len(fns) - 1:
        pool.append(torch.cuda.Stream(device=dev))
    outs = [None] * len(fns)
    for i, fn in enumerate(fns[1:]):
        pool[i].wait_stream(cur)
        with torch.cuda.stream(pool[i]):
            outs[i + 1] = fn()
    outs[0] = fns[0]()
    for i in range(len(fns) - 1):
        cur.wait_stream(pool[i])
    return outs


ASYNC_WGRAD = False      # set for the duration of DLSVAE.backward(): work nothing downstream waits for (weight-gradient
_SIDE = {}               # GEMMs into gradient sinks, the decoder's embedding gradient) leaves the critical path


def _side_state(dev=None):
    cur = torch.cuda.current_stream(dev)
    st = _SIDE.get(cur.device)
    if st is None:
        st = _SIDE[cur.device] = dict(stream=torch.cuda.Stream(device=cur.device), keep=[])
    return cur, st


class side_work(object):
    """`with side_work(enabled, keep=[...])`: run the body on the device's side stream, forked from the current stream.
    In a backward pass (ASYNC_WGRAD) the body may only accumulate into gradient sinks or produce tensors whose consumers
    themselves run on the side stream (forward under on_side): autograd orders a consumer after the stream the node's
    FORWARD ran on, not after this one.  `keep`: tensors of the current stream that the body reads; they stay referenced
    until join_side_work() so that the caching allocator cannot hand their memory out."""

    def __init__(self, enabled, keep=()):
        self.enabled = bool(enabled) and ASYNC_WGRAD
        self.keep = keep

    def __enter__(self):
        if not self.enabled:
            return self
        cur, st = _side_state()
        if cur == st["stream"]:              # already there (a node whose forward ran under on_side)
            self.enabled = False
            return self
        st["stream"].wait_stream(cur)
        st["keep"].append(self.keep)
        self.ctx = torch.cuda.stream(st["stream"])
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
        return False


class on_side(object):
    """Forward-time region on the side stream, joined on exit: the autograd nodes created inside run their BACKWARD on
    the side stream too, i.e. beside whatever the main stream does then (the decoder's input embedding: its gradient --
    a sort and a segmented sum -- is needed by nobody before the optimizer)."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and torch.cuda.is_available()

    def __enter__(self):
        if not self.enabled:
            return self
        self.cur, st = _side_state()
        self.side = st["stream"]
        self.side.wait_stream(self.cur)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
            self.cur.wait_stream(self.side)
        return False


SIDE_DX = os.environ.get("PEVAE_SIDE_DX", "1") == "1"
LEAF_DECODER_INPUT = os.environ.get("PEVAE_LEAF_DECODER_INPUT", "1") == "1"
# set by DLSVAE.forward around the encoder's launch (A/B: PEVAE_HEAD_SIDE): side work of the step's head beside the recurrence
FWD_RNN_BESIDE_SIDE = False
# the decoder's teacher-forced input (embedding, dropout, input projection) made beside the latent stack (A/B switch)
HOIST_DECODER_INPUT = os.environ.get("PEVAE_HOIST_DECODER_INPUT", "1") == "1"
SIDE_BLOCK_WGRAD = os.environ.get("PEVAE_SIDE_BLOCK_WGRAD", "1") == "1"
# > 0: the vocabulary dW runs on that many workgroups on a second stream beside the latent stack's backward.  Measured at
# the Yelp shape: 128 / 160 / 192 workgroups -> 3.94 / 3.83 / 3.77 ms per step against 3.68 with the kernel left in front of
# the decoder recurrence (a workgroup of it owns a whole CU, and the chain of small kernels it runs beside then waits for
# CUs): off by default.
VOCAB_DW_SIDE_WGS = int(os.environ.get("PEVAE_VOCAB_DW_SIDE_WGS", "0"))
# The vocabulary dW in three vocabulary ranges: the first on the whole chip where it stands, the second and third beside the
# decoder's and the encoder's backward recurrence (half the chip each).  "a,b": fractions of the rows given to ranges 2 and 3.
VOCAB_DW_BESIDE = os.environ.get("PEVAE_VOCAB_DW_BESIDE", "0.5,0.5")
VOCAB_DW_BESIDE = tuple(float(x) for x in VOCAB_DW_BESIDE.split(",")) if VOCAB_DW_BESIDE not in ("", "0") else None


def _vocab_dw_cuts(V):
    """-> (c1, c2): ranges [0, c1) now, [c1, c2) and [c2, V) beside the recurrences; multiples of 128 rows (the dW kernel's
    unit).  None when the vocabulary is too small to be worth three launches."""
    if VOCAB_DW_BESIDE is None or V < 4096:
        return None
    f2, f3 = VOCAB_DW_BESIDE
    n2, n3 = int(V * f2) // 128 * 128, int(V * f3) // 128 * 128
    c1 = (V - n2 - n3) // 128 * 128
    if c1 < 0 or n2 <= 0 or n3 <= 0:
        return None
    if c1 < 1024:                # a first range of a few units is a whole-chip launch for nothing: the second takes its rows
        n2, c1 = n2 + c1, 0
    return c1, c1 + n2
_DEFERRED = []           # (event on the stream that queued it, closure): side work whose launch is postponed


def defer_side_work(fn, keep=()):
    """Run fn() on the side stream LATER (at the next join_side_work()), ordered after everything the current stream has
    queued so far.  Postponing the launch changes nothing in what the work depends on, but under HIP-graph capture it
    makes the critical chain, not the side work, the first successor of the node that produced their common input:
    the graph executor keeps a node's first successor on its queue and starts the others through a cross-queue wait."""
    if not ASYNC_WGRAD:
        fn()
        return
    cur, st = _side_state()
    ev = torch.cuda.Event()
    ev.record(cur)
    _DEFERRED.append((ev, fn, keep))


EAGER_FLUSH = os.environ.get("PEVAE_EAGER_FLUSH", "1") == "1"


def flush_deferred():
    """Launch the postponed side work NOW on the side stream, without joining: called by a node of the critical chain right
    after it has queued its own kernels (which are then the first successors of what they share with the side work).  The
    latent stack's backward is a chain of ~10 Block nodes; without this their parameter-gradient GEMMs all waited for the
    join in front of the encoder recurrence and ran there one after another while the main stream idled (~290 us of the
    round-2 step's timeline)."""
    if not _DEFERRED or not EAGER_FLUSH:
        return
    cur, st = _side_state()
    side = st["stream"]
    if cur == side:
        return
    while _DEFERRED:
        ev, fn, keep = _DEFERRED.pop(0)
        st["keep"].append(keep)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            fn()


_AFTER_RNN = []          # closures for the SECOND side stream, launched right after the next persistent recurrence kernel


def defer_after_recurrence(fn, keep=()):
    """Run fn() on a second side stream once the next persistent GRU / LSTM backward kernel has been queued (ordered after
    that kernel, which needs every CU): the vocabulary projection's weight gradient -- a long full-chip kernel nothing
    reads before the optimizer -- then runs beside the latent stack's backward instead of in front of the recurrence.
    Without ASYNC_WGRAD it runs at once."""
    if not ASYNC_WGRAD:
        fn()
        return
    _AFTER_RNN.append((fn, keep))


_BESIDE_RNN = []         # closures fn(wgs) that want to run BESIDE a persistent recurrence kernel, one per recurrence, FIFO


def defer_beside_recurrence(fn, keep=()):
    """fn(wgs) runs on the second side stream, forked right in front of the next persistent GRU / LSTM backward launch, on
    wgs = half the CUs, while that recurrence runs on the other half (its step is a chain of cross-workgroup latencies: on
    half the chip it loses ~10 %, and a full-chip kernel that nothing downstream reads gets the other half for free).
    What is still waiting at the end of the backward pass runs there with wgs = 0 (the whole chip).  Without ASYNC_WGRAD:
    at once, whole chip."""
    if not ASYNC_WGRAD:
        fn(0)
        return
    _BESIDE_RNN.append((fn, keep))


def launch_beside_recurrence():
    """Called by a recurrence's backward right before its launch, after join_side_work().  -> the workgroup budget of the
    recurrence launch (0 = whole chip)."""
    if not _BESIDE_RNN:
        return 0
    cur, st = _side_state()
    s2 = _second_stream(st)
    half = _half_chip(cur.device)
    fn, keep = _BESIDE_RNN.pop(0)
    st["keep"].append(keep)
    s2.wait_stream(cur)
    st["busy2"] = True
    with torch.cuda.stream(s2):
        fn(BESIDE_WGS or half)
    return half


_HALF = {}
BESIDE_WGS = int(os.environ.get("PEVAE_BESIDE_WGS", "0"))      # experiments: workgroups of the kernel beside the recurrence


def _half_chip(dev):
    if dev not in _HALF:
        _HALF[dev] = torch.cuda.get_device_properties(dev).multi_processor_count // 2
    return _HALF[dev]


def _second_stream(st):
    if "stream2" not in st:
        st["stream2"] = torch.cuda.Stream(device=st["stream"].device)
    return st["stream2"]


def launch_after_recurrence():
    """Called right after a persistent recurrence kernel has been queued on the current stream."""
    if not _AFTER_RNN:
        return
    cur, st = _side_state()
    s2 = _second_stream(st)
    s2.wait_stream(cur)
    st["busy2"] = True
    while _AFTER_RNN:
        fn, keep = _AFTER_RNN.pop(0)
        st["keep"].append(keep)
        with torch.cuda.stream(s2):
            fn()


class fork_side(object):
    """Forward-time region on the side stream that is NOT joined on exit: the next join_side_work() (every persistent
    recurrence launch makes one) orders the current stream after it.  For work whose results are first read behind such
    a join: the spectral-norm pre-pass of the whole latent stack runs beside the embedding / projection launches in front
    of the encoder recurrence."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and torch.cuda.is_available()

    def __enter__(self):
        if not self.enabled:
            return self
        cur, st = _side_state()
        if cur == st["stream"]:
            self.enabled = False
            return self
        st["stream"].wait_stream(cur)
        self.ctx = torch.cuda.stream(st["stream"])
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
        return False


class fork_third(object):
    """Forward-time region on a THIRD stream, forked from the current one and joined by join_third() (and by every
    join_side_work() of the step from then on: the region's autograd nodes run their backward there whenever the engine
    reaches them).  For a short branch that must not queue behind what the side stream carries at that point (the
    decoder half's preparation runs there beside the latent stack): DLS_Auto's batched consistency losses."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and torch.cuda.is_available()

    def __enter__(self):
        if not self.enabled:
            return self
        cur, st = _side_state()
        if "stream3" not in st:
            st["stream3"] = torch.cuda.Stream(device=cur.device)
        if cur == st["stream3"]:
            self.enabled = False
            return self
        st["stream3"].wait_stream(cur)
        st["busy3"] = True
        self.ctx = torch.cuda.stream(st["stream3"])
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
        return False


def join_third():
    """The current stream waits for what the third stream holds."""
    if not _SIDE:
        return
    cur, st = _side_state()
    if st.get("busy3") and cur != st["stream3"]:
        cur.wait_stream(st["stream3"])


def third_stream_reset():
    """Head of a forward pass: no join waits for the third stream until a fork_third region has used it in this step
    (a stream that was forked into a HIP-graph capture stays part of it, so waiting for it again is legal and free; one
    that never joined this capture must not be waited for)."""
    if _SIDE:
        _cur, st = _side_state()
        st["busy3"] = False


def join_side_work():
    """Launch postponed side work, then make the current stream wait for the side streams (before a persistent kernel
    that needs every CU, and at the end of the backward pass)."""
    if not _SIDE and not _DEFERRED and not _AFTER_RNN and not _BESIDE_RNN:
        return
    cur, st = _side_state()
    side = st["stream"]
    if cur == side:
        return
    if _BESIDE_RNN and not ASYNC_WGRAD:          # the end of the backward pass: no recurrence came for these
        while _BESIDE_RNN:
            fn, _keep = _BESIDE_RNN.pop(0)
            fn(0)
    while _DEFERRED:
        ev, fn, _keep = _DEFERRED.pop(0)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            fn()
    cur.wait_stream(side)
    if _AFTER_RNN:                     # no recurrence came: run what waited for one now
        launch_after_recurrence()
    if st.get("busy2"):
        cur.wait_stream(st["stream2"])
        st["busy2"] = False
    if st.get("busy3") and cur != st["stream3"]:
        cur.wait_stream(st["stream3"])
    st["keep"] = []


def workspace(tag, numel, dtype, device, zero=False):
    """Persistent scratch (reused every step, stream-ordered: one buffer per tag AND stream, so that the side stream's
    kernels never share scratch with the main stream's).  Never handed to autograd as saved state."""
    key = (tag, dtype, device, torch.cuda.current_stream(device).stream_id if torch.cuda.is_available() else 0)
    t = _WS.get(key)
    if t is None or t.numel() < numel:
        t = torch.zeros(numel, dtype=dtype, device=device) if zero else torch.empty(numel, dtype=dtype, device=device)
        _WS[key] = t
    return t[:numel]


def device_info():
    cu, lds = C.c_int(0), C.c_int(0)
    arch = C.create_string_buffer(64)
    _lib.call("pv_device_info", C.byref(cu), C.byref(lds), arch, 64)
    return dict(cu_count=cu.value, lds_bytes=lds.value, arch=arch.value.decode())


def gru_geometry(B, H):
    hk, k3, bp, ng = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.call("pv_gru_geometry", B, H, C.byref(hk), C.byref(k3), C.byref(bp), C.byref(ng))
    return hk.value, k3.value, bp.value, ng.value


def lstm_geometry(B, H):
    hk, k4, bp, ng = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.call("pv_lstm_geometry", B, H, C.byref(hk), C.byref(k4), C.byref(bp), C.byref(ng))
    return hk.value, k4.value, bp.value, ng.value


# ------------------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------------------

def cast_pack(src, n0, n1, n1p, ncol, s0, s1, sc, ldd, out=None):
    """bf16 [n0*n1p][ldd] with out[(i0*n1p+i1), c] = src[i0*s0 + i1*s1 + c*sc] (zero K padding)."""
    _need_cuda(src)
    if out is None:
        out = (torch.zeros if n1p != n1 else torch.empty)((n0 * n1p, ldd), dtype=BF16, device=src.device)
    _lib.call("pv_cast_pack_bf16", _p(src), n0, n1, n1p, ncol, s0, s1, sc, _p(out), ldd, _stream())
    return out


_PACK_CACHE = None      # see packed_weights()


class packed_weights(object):
    """Scope in which the bf16 operand copy of a weight matrix is made once and reused: a free-running decode calls the
    same projection and recurrence weights once per token, and nothing updates them in between (decoders.py:243-262 runs
    under the caller's no-grad evaluation or between two optimizer steps).  Never keep it open across an optimizer step."""

    def __enter__(self):
        global _PACK_CACHE
        self.prev, _PACK_CACHE = _PACK_CACHE, ({} if _PACK_CACHE is None else _PACK_CACHE)
        return self

    def __exit__(self, *exc):
        global _PACK_CACHE
        _PACK_CACHE = self.prev
        return False


_PREPACK = {}           # (data_ptr, shape, ld) -> bf16 operand made ahead of its use on the side stream (prepack_weights)
_PRECOMPACT = {}        # see precompact_targets()


_PREPROJ = {}           # see preproject_gru_input()


_PRESORTED = {}         # see presort_lookup()
PRESORT_SIDE = os.environ.get("PEVAE_PRESORT_SIDE", "1") == "1"     # A/B switch


def clear_prepared():
    _PREPACK.clear()
    _PRECOMPACT.clear()
    _PREPROJ.clear()
    _PRESORTED.clear()
    _PRETRANS.clear()


def presort_lookup(ids, weight, tag):
    """The stable (row id, occurrence) sort of a lookup's table gradient, made NOW -- the ids are inputs of the step, the
    gradient rows arrive at the tail of the backward pass, where the sort otherwise stands in a serial chain of small
    kernels in front of the optimizer.  The caller runs this under fork_side() (DLSVAE._prepare_on_side: beside the latent
    stack, joined in front of the decoder), so the backward pass needs no event of its own: embedding_backward() finds the
    entry by the ids' storage and is left with the segmented sum.  One persistent workspace per `tag` (a lookup site)."""
    rows, D = weight.shape
    if not PRESORT_SIDE or grad_sink(weight, (rows, D)) is None:
        return
    ids_c = ids.contiguous()
    n = ids_c.numel()
    if not n:
        return
    nbytes = _lib.load().pv_embedding_bwd_ws_bytes(n, D)
    ws = workspace("emb_presort_" + tag, nbytes, torch.uint8, weight.device)
    _lib.call("pv_embedding_sort", _p(ids_c), n, n, n, 0, 1, rows, D, _p(ws), nbytes, _stream())
    _PRESORTED[(ids_c.data_ptr(), ids_c._version, n, rows, D)] = (ws, nbytes, ids_c)     # (ids_c held: its storage stays its own)


def preproject_gru_input(x, w_ih, b_ih, H):
    """The part of a GRU layer's forward that does not depend on its initial state -- the bf16 operand copy of x [B,L,E] and
    the input projection x W_ih^T + b_ih -- made NOW (the caller runs this under fork_side(), beside the latent stack) for
    GRULayer.forward to pick up ONCE later in the same pass, behind a join_side_work(): the decoder's teacher-forced input
    is known at the start of the step, and its 40 us of gather / cast / GEMM launches otherwise stand between the latent
    stack and the decoder recurrence."""
    B, L, E = x.shape
    Hk, K3, Bpad, NG = gru_geometry(B, H)
    Ep = round_up(E, 64)
    x_bf = cast_pack(x, L, B, Bpad, E, x.stride(1), x.stride(0), x.stride(2), Ep)
    wih_bf = pack_matrix(w_ih, Ep)
    xprojT = gemm_nt_ct(x_bf, Ep, wih_bf, Ep, L * Bpad, 3 * H, Ep, b_ih.contiguous(),
                        workspace("gru_xprojT", 3 * H * L * Bpad, torch.float32, x.device).view(3 * H, L * Bpad))
    _PREPROJ[(x.data_ptr(), tuple(x.shape), w_ih.data_ptr())] = (x_bf, xprojT)


def prepack_weights(pairs):
    """Make the bf16 operand copies of weight matrices now -- the caller runs this under fork_side() at the start of a
    forward pass, beside the first launches -- for pack_matrix() to hand out ONCE later in the same pass (the projection
    weight alone is a 61 MB pass in front of the vocabulary kernels otherwise).  The consumer must sit behind a
    join_side_work(); entries that nobody asked for are dropped by the next clear_prepared()."""
    for w, ld in pairs:
        R, Cc = w.shape
        _PREPACK[(w.data_ptr(), R, Cc, tuple(w.stride()), ld)] = cast_pack(w.contiguous(), 1, R, R, Cc, 0, Cc, 1, ld)


def precompact_targets(labels, layout, ignore_index):
    """The vocabulary kernels' target vector in operand row order (time-major, padded to Bpad rows per step, with
    `layout` = (L, B, Bpad); else flat) and the stable partition of its live rows (pv_vocab_compact): both depend on the
    labels alone, so they are made at the start of the forward pass on the side stream.  -> key for take_precompact()."""
    B, L = labels.shape[0], labels.shape[1]
    if layout is not None:
        Lh, Bh, Bpad = layout
        tgt = labels.reshape(B, L).t()
        if Bpad != B:
            tgt = torch.cat([tgt, torch.full((L, Bpad - B), ignore_index, dtype=labels.dtype, device=labels.device)], dim=1)
        tgt = tgt.contiguous().view(-1)
    else:
        tgt = labels.reshape(-1).contiguous()
    T = tgt.numel()
    perm = count = None
    if SKIP_IGNORED_TOKENS and ignore_index >= 0:
        perm = torch.empty(T, dtype=torch.int64, device=tgt.device)
        count = torch.empty(1, dtype=torch.int32, device=tgt.device)
        _lib.call("pv_vocab_compact", _p(tgt), T, ignore_index, _p(perm), _p(count), _stream())
    _PRECOMPACT[(labels.data_ptr(), labels._version, tuple(labels.shape), layout, ignore_index)] = (tgt, perm, count)


def take_precompact(labels, layout, ignore_index):
    return _PRECOMPACT.pop((labels.data_ptr(), labels._version, tuple(labels.shape), layout, ignore_index), None)


def pack_matrix(w, ld=None):
    """fp32 [R,C] contiguous -> bf16 [R][ld] (ld = round_up(C,64) by default)."""
    R, Cc = w.shape
    ld = round_up(Cc, 64) if ld is None else ld
    if _PREPACK:
        hit = _PREPACK.pop((w.data_ptr(), R, Cc, tuple(w.stride()), ld), None)
        if hit is not None:
            return hit
    if _PACK_CACHE is not None:
        key = (w.data_ptr(), R, Cc, tuple(w.stride()), ld)
        hit = _PACK_CACHE.get(key)
        if hit is None:
            hit = _PACK_CACHE[key] = cast_pack(w.contiguous(), 1, R, R, Cc, 0, Cc, 1, ld)
        return hit
    return cast_pack(w.contiguous(), 1, R, R, Cc, 0, Cc, 1, ld)


_PRETRANS = {}          # (data_ptr, R, Cc, lds, rpad) -> transposed bf16 weight made ahead (pretranspose_weights)
PRETRANSPOSE = os.environ.get("PEVAE_PRETRANSPOSE", "1") == "1"     # A/B switch


def pretranspose_weights(items):
    """The bf16 transposes of recurrence weights that the BACKWARD pass multiplies by (W_hh^T for the recurrence kernel, W_ih^T
    for the input gradient), made during the forward pass on the side stream: the weights do not change in between, and
    two of the four 5 us launches otherwise stand in front of a backward recurrence / the embedding gradient at the tail.
    items: (w, R, Cc, lds, rpad) as transpose_to_bf16 takes them; consumed once each."""
    if not PRETRANSPOSE:
        return
    for w, R, Cc, lds, rpad in items:
        wc = w.contiguous()
        _PRETRANS[(wc.data_ptr(), R, Cc, lds, rpad)] = (transpose_to_bf16(wc, R, Cc, lds, rpad, ahead=True), wc)


def transpose_to_bf16(src, R, Cc, lds, rpad, out=None, ahead=False):
    """bf16 [Cc][rpad] = src[R][lds]^T (src fp32 or bf16), columns R..rpad zero."""
    _need_cuda(src)
    if _PRETRANS and out is None and not ahead:
        hit = _PRETRANS.pop((src.data_ptr(), R, Cc, lds, rpad), None)
        if hit is not None:
            return hit[0]
    if out is None:
        out = torch.empty((Cc, rpad), dtype=BF16, device=src.device)
    _lib.call("pv_transpose_to_bf16", _p(src), 1 if src.dtype == torch.float32 else 0, R, Cc, lds, _p(out), rpad,
              rpad, _stream())
    return out


def gemm_nt(A, lda, Bm, ldb, M, N, K, bias=None, out=None, out_bf16=None, accumulate=False, bias_per_row=False,
            m_live=None, k_live=None):
    """C[M,N] (+)= A[M,K] B[N,K]^T (+bias) with bf16 operands on the MFMA core.  m_live / k_live: int32 device scalars
    bounding the rows / the K extent at run time (pv_gemm_bf16_nt)."""
    _need_cuda(A, Bm)
    if out is None and out_bf16 is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _lib.call("pv_gemm_bf16_nt", _p(A), lda, _p(Bm), ldb, M, N, K, _p(bias), 1 if bias_per_row else 0, _p(out),
              0 if out is None else out.stride(0), _p(out_bf16), 0 if out_bf16 is None else out_bf16.stride(0),
              1 if accumulate else 0, _p(m_live), _p(k_live), _stream())
    return out if out is not None else out_bf16


def gemm_nt_ct(A, lda, Bm, ldb, M, N, K, bias, out_t):
    """out_t[N, M] = (A[M,K] B[N,K]^T + bias[N])^T in fp32, stored with 16-byte stores (pv_gemm_bf16_nt_ct)."""
    _need_cuda(A, Bm, out_t)
    _lib.call("pv_gemm_bf16_nt_ct", _p(A), lda, _p(Bm), ldb, M, N, K, _p(bias), _p(out_t), out_t.stride(0), _stream())
    return out_t


def gemm_nt_wgrad(A, lda, Bm, ldb, M, N, K, sink=None, k_live=None, ws_tag="gemm_splitk"):
    """Weight-gradient GEMM C[M,N] = A[M,K] B[N,K]^T whose K is the token count: split-K over ~2 workgroups per CU when
    the plain tiling would leave most of the chip idle.  Accumulates into `sink` when given, else returns C."""
    _need_cuda(A, Bm)
    cus = device_info()["cu_count"] if not _DEV_CU else _DEV_CU[0]
    if not _DEV_CU:
        _DEV_CU.append(cus)
    splits = C.c_int(1)
    _lib.call("pv_gemm_splitk_plan", M, N, K, 2 * cus, C.byref(splits))
    out = sink if sink is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    if splits.value <= 1:
        gemm_nt(A, lda, Bm, ldb, M, N, K, out=out, accumulate=sink is not None, k_live=k_live)
    else:
        ws = workspace(ws_tag, _lib.load().pv_gemm_splitk_ws_bytes(M, N, splits.value), torch.uint8, A.device)
        _lib.call("pv_gemm_bf16_nt_splitk", _p(A), lda, _p(Bm), ldb, M, N, K, splits.value, _p(ws), _p(out),
                  out.stride(0), 1 if sink is not None else 0, None, _p(k_live), _stream())
    return None if sink is not None else out


GEMM_TN = os.environ.get("PEVAE_GEMM_TN", "1") == "1"      # A/B switch: weight gradients without transposed operand copies


def gemm_tn_wgrad(A, lda, Bm, ldb, M, N, K, sink=None, ws_tag="gemm_splitk"):
    """Weight-gradient GEMM C[M,N] = A^T B over K-major operands A [K][lda], B [K][ldb] (bf16; K = the token count, a
    multiple of 64): what gemm_nt_wgrad computes from transposed copies, without the copies (pv_gemm_bf16_tn_splitk).
    Accumulates into `sink` when given, else returns C."""
    _need_cuda(A, Bm)
    cus = device_info()["cu_count"] if not _DEV_CU else _DEV_CU[0]
    if not _DEV_CU:
        _DEV_CU.append(cus)
    splits = C.c_int(1)
    _lib.call("pv_gemm_splitk_plan", M, N, K, 2 * cus, C.byref(splits))
    out = sink if sink is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    ws = None
    if splits.value > 1:
        ws = workspace(ws_tag, _lib.load().pv_gemm_splitk_ws_bytes(M, N, splits.value), torch.uint8, A.device)
    _lib.call("pv_gemm_bf16_tn_splitk", _p(A), lda, _p(Bm), ldb, M, N, K, max(1, splits.value), _p(ws), _p(out),
              out.stride(0), 1 if sink is not None else 0, _stream())
    return None if sink is not None else out


def gemm_nt_live_rows(A, lda, Bm, ldb, M, N, K, m_live, max_splits=4):
    """C[M,N] = A[M,K] B[N,K]^T where only the first *m_live rows (int32 device scalar) are wanted: the kernel walks the
    live tiles only and splits K on the device so that they still fill the chip (pv_gemm_bf16_nt_splitk with m_live).
    Rows past round_up(*m_live, 128) of the result are uninitialised."""
    _need_cuda(A, Bm)
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    splits = max(1, min(max_splits, K // 64 // 4))
    ws = workspace("gemm_splitk_rows", _lib.load().pv_gemm_splitk_ws_bytes(M, N, splits), torch.uint8, A.device)
    _lib.call("pv_gemm_bf16_nt_splitk", _p(A), lda, _p(Bm), ldb, M, N, K, splits, _p(ws), _p(out), out.stride(0), 0,
              _p(m_live), None, _stream())
    return out


_DEV_CU = []


def grad_sink(param, shape=None):
    """A parameter that already owns a dense fp32 `.grad` (FlatAdam's views into the flat gradient buffer, or
    any earlier backward) lets the backward kernels ACCUMULATE into that buffer in place -- the autograd node
    then returns None for it, which removes the temporary, its memset and the AccumulateGrad add kernel.
    Parameters without a `.grad` get their gradient returned the ordinary way."""
    g = getattr(param, "grad", None)
    if (isinstance(param, torch.nn.Parameter) and g is not None and g.is_cuda and g.dtype == torch.float32
            and g.is_contiguous() and (shape is None or tuple(g.shape) == tuple(shape))):
        return g
    return None


def masked_sum(x, mask=None, ignore=0):
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.call("pv_masked_sum", _p(x), _p(mask), ignore, x.numel(), _p(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------------------
# dropout (nn.Dropout of encoders.py:54, decoders.py:175,189) on our own counter-based generator
# ------------------------------------------------------------------------------------------------------------

_RNG = {}


def rng_state(device):
    """{uint64 seed, uint64 step} in device memory (csrc/dropout.hip); seeded on first use from torch.initial_seed() and the
    data-parallel rank (replicas seed torch identically to start from identical weights: without the rank every replica
    would draw the same masks for its different samples).  The masks of a step are regenerated in backward from (seed, step,
    site): ONE training-mode forward per backward -- a second forward before the first one's backward advances the step
    and the first backward would apply other masks."""
    st = _RNG.get(device)
    if st is None:
        import struct
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        raw = struct.pack("QQ", (torch.initial_seed() + 0x9E3779B97F4A7C15 * rank) & 0xFFFFFFFFFFFFFFFF, 0)
        st = _RNG[device] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    return st


def rng_seed(seed, device):
    import struct
    rng_state(device).copy_(torch.frombuffer(bytearray(struct.pack("QQ", seed & 0xFFFFFFFFFFFFFFFF, 0)), dtype=torch.uint8))


def rng_advance(device):
    """step += 1 on the device: once per training step (DLSVAE.forward), so that a replayed HIP graph draws new masks."""
    _lib.call("pv_rng_advance", _p(rng_state(device)), _stream())


def _drop_call(x3, out3, p, site, mask, midx):
    n0, n1, Cc = x3.shape
    _lib.call("pv_dropout_f32", _p(x3), n0, n1, Cc, x3.stride(0), x3.stride(1), x3.stride(2), midx[0], midx[1], float(p),
              _p(mask), _p(rng_state(x3.device)), int(site), _p(out3), out3.stride(0), out3.stride(1), _stream())


class DropoutFn(torch.autograd.Function):
    """y = x * keep / (1 - p); keep from `mask` (uint8, same shape as x) or regenerated from (seed, step, site, index) in
    both passes -- no mask is stored."""

    @staticmethod
    def forward(ctx, x, p, site, mask):
        _need_cuda(x, mask)
        xc = x.contiguous()
        Cc = xc.shape[-1]
        x3 = xc.view(1, -1, Cc)
        out = torch.empty_like(xc)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
            assert mask.numel() == xc.numel()
        _drop_call(x3, out.view(1, -1, Cc), p, site, mask, (0, Cc))
        ctx.p, ctx.site, ctx.mask = p, site, mask
        return out

    @staticmethod
    def backward(ctx, dy):
        dyc = dy.contiguous()
        Cc = dyc.shape[-1]
        dx = torch.empty_like(dyc)
        _drop_call(dyc.view(1, -1, Cc), dx.view(1, -1, Cc), ctx.p, ctx.site, ctx.mask, (0, Cc))
        return dx, None, None, None


_SITE = [0]


class Dropout(torch.nn.Module):
    """torch.nn.Dropout's interface on the kernels above.  `mask` (uint8 / bool tensor of the input's shape, 1 = keep)
    overrides the generator: parity tests hand the same mask to the oracle.  Every forward call draws under its own
    site id (instance id x 4096 + call number), so a module called once per decoded token does not repeat its mask."""

    def __init__(self, p=0.5):
        super().__init__()
        if p < 0 or p >= 1:
            if p == 1:
                raise ValueError("pevae_amd Dropout: p = 1 is not supported (every element dropped)")
            raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(p))
        self.p = float(p)
        _SITE[0] += 1
        self.site_base, self.calls, self.mask = _SITE[0] * 4096, 0, None

    def next_site(self):
        self.calls = (self.calls + 1) % 4096
        return self.site_base + self.calls

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return DropoutFn.apply(x, self.p, self.next_site(), self.mask)

    def extra_repr(self):
        return "p={}".format(self.p)


# ------------------------------------------------------------------------------------------------------------
# fused vocabulary projection + log-softmax + NLL
# ------------------------------------------------------------------------------------------------------------

def _vocab_operands(hid, weight):
    T, H = hid.shape
    V = weight.shape[0]
    Hp = round_up(H, 64)
    if hid.dtype == BF16:
        assert hid.stride(1) == 1 and hid.stride(0) % 8 == 0 and hid.stride(0) >= Hp
        h_bf, ldh = hid, hid.stride(0)
    else:
        h_bf, ldh = cast_pack(hid.contiguous(), 1, T, T, H, 0, H, 1, Hp), Hp
    w_bf = pack_matrix(weight, Hp)
    return T, H, V, Hp, h_bf, ldh, w_bf


SKIP_IGNORED_TOKENS = os.environ.get("PEVAE_VOCAB_DENSE", "0") != "1"     # see VocabNLL
# H = 512 with gradients wanted: the two flash-attention-shaped kernels of csrc/vocab_fused.hip (no [T, V] matrix in HBM in
# either pass) instead of the four tile GEMMs with the P / P^T round trip.  PEVAE_VOCAB_FUSED=0 keeps the tile-GEMM path.
FUSED_VOCAB = os.environ.get("PEVAE_VOCAB_FUSED", "1") != "0"


FUSE_DROP_SCATTER = os.environ.get("PEVAE_FUSE_DROP_SCATTER", "1") == "1"


class VocabNLL(torch.autograd.Function):
    """nll_tok[t] = -log_softmax((h W^T + b) / temperature)[target[t]], 0 for ignored targets; the [T,V]
    log-prob tensor is never built (decoders.py:191-193 + criterions.py:55-64 of the reference, fused).

    Rows whose target is the ignore index add exactly zero to the loss and to every gradient (NLLLoss(ignore_index),
    criterions.py:38-41), so by default they are not multiplied at all: a stable partition moves the live rows to the
    front (pv_vocab_compact / pv_vocab_gather) and the four GEMMs of the group take the live-row count from device
    memory (static launch geometry, graph-capturable).  PEVAE_VOCAB_DENSE=1 computes every row like the reference."""

    RETURNED_SUMS = False       # set by every forward: True when `nll` is the {sum, live count} pair asked for with sums_only

    @staticmethod
    def forward(ctx, hid, weight, bias, target, inv_temp, ignore_index, hid_bf16, layout, drop=None, pre=None, sums_only=False):
        # hid: fp32 [T,H] -- or, with `hid_bf16` ([L*Bpad, ld] time-major bf16 rows from the GRU kernel) and
        # layout=(L,B,Bpad), the batch-first [B,L,H] tensor the gradient is returned for.  `target` follows
        # the row order of the bf16 operand.
        # drop = (p, site, mask): decoders.py:189 output dropout folded into the operand pack -- hid [B,L,H] fp32 (any
        # strides) is dropped, scaled and written as the bf16 operand in one pass (time-major rows with `layout`, else
        # batch-major rows b*L + t); the backward pass applies the same keep decisions to dH.
        _need_cuda(hid, weight, bias, target)
        ctx.drop = None
        if drop is not None:
            p_, site_, mask_ = drop
            Bh, Lh, Hh = hid.shape
            Hp_ = round_up(Hh, 64)
            if mask_ is not None:
                mask_ = mask_.to(torch.uint8).contiguous()
                assert mask_.numel() == Bh * Lh * Hh
            if layout is not None:
                L_, B_, Bpad_ = layout
                n0, n1, n1p, s0, s1, m0, m1 = L_, B_, Bpad_, hid.stride(1), hid.stride(0), Hh, Lh * Hh
            else:
                n0, n1, n1p, s0, s1, m0, m1 = Bh, Lh, Lh, hid.stride(0), hid.stride(1), Lh * Hh, Hh
            hid_bf16 = torch.empty((n0 * n1p, Hp_), dtype=BF16, device=hid.device)
            _lib.call("pv_dropout_pack_bf16", _p(hid), n0, n1, n1p, Hh, s0, s1, hid.stride(2), m0, m1, float(p_), _p(mask_),
                      _p(rng_state(hid.device)), int(site_), _p(hid_bf16), Hp_, _stream())
            ctx.drop = (float(p_), int(site_), mask_, (n0, n1, n1p, m0, m1))
        T, H, V, Hp, h_bf, ldh, w_bf = _vocab_operands(hid if hid_bf16 is None else hid_bf16[:, :weight.shape[1]], weight)
        dev = weight.device
        ctx.layout = layout
        target = target.contiguous()
        perm = count = None
        ctx.fused = False
        VocabNLL.RETURNED_SUMS = False
        if FUSED_VOCAB and any(ctx.needs_input_grad[:3]) and _lib.load().pv_vocab_fused_supported(Hp):
            return VocabNLL._forward_fused(ctx, hid, weight, bias, target, inv_temp, ignore_index, T, H, V, Hp, h_bf, ldh, w_bf,
                                           pre, sums_only)
        if SKIP_IGNORED_TOKENS and ignore_index >= 0:
            if pre is not None and pre[0] is not None:
                perm, count = pre
            else:
                perm = torch.empty(T, dtype=torch.int64, device=dev)
                count = torch.empty(1, dtype=torch.int32, device=dev)
                _lib.call("pv_vocab_compact", _p(target), T, ignore_index, _p(perm), _p(count), _stream())
            h_c = torch.empty((T, Hp), dtype=BF16, device=dev)
            t_c = torch.empty(T, dtype=torch.int64, device=dev)
            _lib.call("pv_vocab_gather", _p(h_bf), ldh, _p(target), _p(perm), T, Hp, _p(h_c), Hp, _p(t_c), _stream())
            h_bf, ldh, target = h_c, Hp, t_c
        ws = workspace("vocab_fwd", _lib.load().pv_vocab_nll_fwd_ws_bytes(T, V), torch.uint8, dev)
        lse = torch.empty(T, dtype=torch.float32, device=dev)
        nll = torch.empty(T, dtype=torch.float32, device=dev)
        bias_c = bias.contiguous()
        with timed("vocab_fwd"):
            _lib.call("pv_vocab_nll_fwd", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias_c), _p(target), T, V, Hp, inv_temp,
                      ignore_index, _p(ws), _p(lse), _p(nll), _p(count), _stream())
        if perm is not None:                           # back to the caller's row order (ignored rows: 0)
            nll_c, lse_c = nll, lse
            nll = torch.empty(T, dtype=torch.float32, device=dev)
            lse = torch.empty(T, dtype=torch.float32, device=dev)
            _lib.call("pv_scatter_rows_f32", _p(nll_c), 1, _p(perm), _p(count), T, 1, _p(nll), 1, _stream())
            _lib.call("pv_scatter_rows_f32", _p(lse_c), 1, _p(perm), _p(count), T, 1, _p(lse), 1, _stream())
            ctx.save_for_backward(h_bf, w_bf, bias_c, target, lse_c, perm, count)
        else:
            ctx.save_for_backward(h_bf, w_bf, bias_c, target, lse)
        ctx.compact = perm is not None
        ctx.dims = (T, H, V, Hp, ldh, inv_temp, ignore_index)
        ctx.hid_dtype = hid.dtype
        ctx.params = (weight, bias)
        ctx.mark_non_differentiable(lse)
        ctx.set_materialize_grads(False)
        return nll, lse

    @staticmethod
    def _forward_fused(ctx, hid, weight, bias, target, inv_temp, ignore_index, T, H, V, Hp, h_bf, ldh, w_bf, pre=None,
                       sums_only=False):
        dev = weight.device
        perm = count = t_c = None
        t32 = torch.empty(T, dtype=torch.int32, device=dev)
        if SKIP_IGNORED_TOKENS and ignore_index >= 0:
            if pre is not None and pre[0] is not None:      # (perm, count) made ahead by precompact_targets()
                perm, count = pre
            else:
                perm = torch.empty(T, dtype=torch.int64, device=dev)
                count = torch.empty(1, dtype=torch.int32, device=dev)
                _lib.call("pv_vocab_compact", _p(target), T, ignore_index, _p(perm), _p(count), _stream())
            h_c = torch.empty((T, Hp), dtype=BF16, device=dev)
            t_c = torch.empty(T, dtype=torch.int64, device=dev)
            _lib.call("pv_vocab_gather32", _p(h_bf), ldh, _p(target), _p(perm), T, Hp, _p(h_c), Hp, _p(t_c), _p(t32), _stream())
            h_bf, ldh, target = h_c, Hp, t_c
        else:
            _lib.call("pv_vocab_gather32", None, 0, _p(target), None, T, Hp, None, 0, None, _p(t32), _stream())
        ws = workspace("vocab_fused_fwd", _lib.load().pv_vocab_fused_fwd_ws_bytes(T, V), torch.uint8, dev)
        lse = torch.empty(T, dtype=torch.float32, device=dev)
        nll = torch.empty(T, dtype=torch.float32, device=dev)
        U = torch.empty((T, Hp), dtype=torch.float32, device=dev)
        bias_c = bias.contiguous()
        with timed("vocab_fused_fwd"):
            _lib.call("pv_vocab_fused_fwd", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias_c), _p(t32), T, V, Hp, inv_temp, ignore_index,
                      _p(ws), _p(lse), _p(nll), _p(U), _p(count), _stream())
        lse_c = lse
        ctx.sums_only = False
        VocabNLL.RETURNED_SUMS = False
        if perm is not None and sums_only:
            # the caller wants {sum, live count} only: reduce the compact rows where they are (rows past the live count
            # carry the ignore index in the compact target vector) -- no scatter back to the caller's row order
            nll = masked_sum(nll, target, ignore_index)
            ctx.sums_only = True
            VocabNLL.RETURNED_SUMS = True            # (what the caller must look at: the shape of `nll` does not tell)
        elif perm is not None:
            nll_c = nll
            nll = torch.empty(T, dtype=torch.float32, device=dev)
            lse = torch.empty(T, dtype=torch.float32, device=dev)
            _lib.call("pv_scatter_rows_f32", _p(nll_c), 1, _p(perm), _p(count), T, 1, _p(nll), 1, _stream())
            _lib.call("pv_scatter_rows_f32", _p(lse_c), 1, _p(perm), _p(count), T, 1, _p(lse), 1, _stream())
        ctx.fused = True
        ctx.fused_saved = (h_bf, w_bf, bias_c, target, t32, lse_c, perm, count, U)     # plain references: none needs grad
        ctx.dims = (T, H, V, Hp, ldh, inv_temp, ignore_index)
        ctx.params = (weight, bias)
        ctx.mark_non_differentiable(lse)
        ctx.set_materialize_grads(False)
        return nll, lse

    @staticmethod
    def _backward_fused(ctx, dnll):
        if ctx.fused_saved is None:
            raise RuntimeError("VocabNLL: second backward through the fused vocabulary node -- its saved operands (the "
                               "compact states, U = dNLL/dh) are released by the first one; retain_graph is not supported here")
        h_bf, w_bf, bias, target, t32, lse, perm, count, U = ctx.fused_saved
        ctx.fused_saved = None
        T, H, V, Hp, ldh, inv_temp, ignore_index = ctx.dims
        dev = w_bf.device
        gscale = torch.empty(T, dtype=torch.float32, device=dev)      # `target` is already in compact order
        if ctx.sums_only:            # dnll = d{sum, count}: every live row's upstream gradient is d sum
            _lib.call("pv_masked_sum_bwd", _p(target), ignore_index, _p(dnll.to(torch.float32).contiguous()), T, _p(gscale),
                      _stream())
        else:
            _lib.call("pv_vocab_gscale", _p(dnll.to(torch.float32).contiguous()), _p(perm), _p(target), ignore_index, T,
                      _p(gscale), _stream())
        dH = torch.empty((T, H), dtype=torch.float32, device=dev)
        if ctx.drop is not None and FUSE_DROP_SCATTER and T % ctx.drop[3][2] == 0:
            # the output dropout's backward rides in the scatter's store pass (one pass over dH instead of two)
            p_, site_, mask_, (n0, n1, n1p, m0, m1) = ctx.drop
            _lib.call("pv_scatter_rows_scaled_drop_f32", _p(U), _p(gscale), inv_temp, _p(perm), _p(count), T, H, _p(dH), H,
                      n1, n1p, m0, m1, p_, _p(mask_), _p(rng_state(dev)), site_, _stream())
            ctx.drop_done = True
        else:
            _lib.call("pv_scatter_rows_scaled_f32", _p(U), _p(gscale), inv_temp, _p(perm), _p(count), T, H, _p(dH), H, _stream())
        weight, _bias = ctx.params
        sink = grad_sink(weight, (V, H))
        dW = sink if sink is not None else torch.empty((V, H), dtype=torch.float32, device=dev)
        dbias = torch.empty(V, dtype=torch.float32, device=dev)
        sink_b = grad_sink(_bias, (V,)) if sink is not None else None
        if sink_b is not None:
            dbias = sink_b
        ws_bytes = _lib.load().pv_vocab_fused_dw_ws_bytes(V)

        def launch(wgs=0, v0=0, v1=V, tag="vocab_fused_dw"):
            # rows [v0, v1) of the vocabulary (pv_vocab_fused_dw_range; the whole vocabulary: the same call)
            ws = workspace(tag, _lib.load().pv_vocab_fused_dw_ws_bytes(v1 - v0), torch.uint8, dev)     # (of the stream it runs on)
            with timed("vocab_fused_dw"):
                _lib.call("pv_vocab_fused_dw_range", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias), _p(t32), _p(lse), _p(gscale), T, V, Hp,
                          inv_temp, v0, v1, _p(ws), _p(dW), H, 1 if sink is not None else 0, _p(dbias),
                          1 if sink_b is not None else 0, _p(count), wgs, _stream())

        cuts = _vocab_dw_cuts(V) if (sink_b is not None and ASYNC_WGRAD and VOCAB_DW_BESIDE and dW.is_contiguous()
                                      and bias.is_contiguous() and dbias.is_contiguous()) else None
        if cuts is not None:
            # Both gradients sink into the flat buffer and nothing downstream reads them before the optimizer.  The first
            # rows run here on the whole chip; the other two ranges wait for the decoder's and the encoder's recurrence and
            # run beside them, half the chip each (defer_beside_recurrence).
            c1, c2 = cuts
            if c1 > 0:
                launch(0, 0, c1)
            for k, (a, b) in enumerate(((c1, c2), (c2, V))):
                held = (h_bf, w_bf, bias, t32, lse, gscale, count, dW, dbias)
                defer_beside_recurrence(lambda wgs, a=a, b=b, k=k: launch(wgs, a, b, "vocab_fused_dw_r%d" % k), keep=held)
        elif sink_b is not None and ASYNC_WGRAD and VOCAB_DW_SIDE_WGS > 0:
            # both gradients sink into the flat buffer and nothing downstream reads them: off the critical path, on fewer
            # workgroups than CUs so that the latent stack's backward keeps the CUs it needs
            defer_after_recurrence(lambda: launch(VOCAB_DW_SIDE_WGS), keep=(h_bf, w_bf, bias, t32, lse, gscale, count, dW, dbias))
        else:
            launch()
        return dH, (None if sink is not None else dW), (None if sink_b is not None else dbias)

    @staticmethod
    def backward(ctx, dnll, _dlse):
        if dnll is None:
            return (None,) * 11
        if ctx.fused:
            dH, dW, dbias = VocabNLL._backward_fused(ctx, dnll)
            T, H, V, Hp, ldh, inv_temp, ignore_index = ctx.dims
            dev = dH.device
            return VocabNLL._finish_dh(ctx, dH, H, dev) + (dW, dbias, None, None, None, None, None, None, None, None)
        if ctx.compact:
            h_bf, w_bf, bias, target, lse, perm, count = ctx.saved_tensors
        else:
            h_bf, w_bf, bias, target, lse = ctx.saved_tensors
            perm = count = None
        T, H, V, Hp, ldh, inv_temp, ignore_index = ctx.dims
        dev = w_bf.device
        Tp, Vp = round_up(T, 64), round_up(V, 64)
        gscale = torch.empty(T, dtype=torch.float32, device=dev)      # `target` is already in compact order
        _lib.call("pv_vocab_gscale", _p(dnll.to(torch.float32).contiguous()), _p(perm), _p(target), ignore_index, T, _p(gscale),
                  _stream())
        P = workspace("vocab_P", Tp * Vp, BF16, dev)
        PT = workspace("vocab_PT", Vp * Tp, BF16, dev)
        dbp = workspace("vocab_dbias_part", (Tp // 64) * Vp, torch.float32, dev)
        dbias = torch.empty(V, dtype=torch.float32, device=dev)
        with timed("vocab_bwd_prob"):
            _lib.call("pv_vocab_nll_bwd_prob", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias), _p(target), _p(lse),
                      _p(gscale), T, V, Hp, inv_temp, _p(P), Vp, _p(PT), Tp, _p(dbp), _p(dbias), _p(count), _stream())
        weight, _bias = ctx.params
        dH, dW = _vocab_stage2(P, PT, h_bf, ldh, w_bf, T, H, V, Hp, Tp, Vp, grad_sink(weight, (V, H)), count)
        if perm is not None:
            dH_c = dH
            dH = torch.empty((T, H), dtype=torch.float32, device=dev)
            _lib.call("pv_scatter_rows_f32", _p(dH_c), H, _p(perm), _p(count), T, H, _p(dH), H, _stream())
        return VocabNLL._finish_dh(ctx, dH, H, dev) + (dW, dbias, None, None, None, None, None, None, None, None)

    @staticmethod
    def _finish_dh(ctx, dH, H, dev):
        if ctx.drop is not None:                       # dL/d(undropped states) = dH * keep / (1 - p), in place
            p_, site_, mask_, (n0, n1, n1p, m0, m1) = ctx.drop
            if not getattr(ctx, "drop_done", False):   # (the fused backward folds it into its scatter pass)
                _lib.call("pv_dropout_f32", _p(dH), n0, n1, H, n1p * H, H, 1, m0, m1, p_, _p(mask_), _p(rng_state(dev)), site_,
                          _p(dH), n1p * H, H, _stream())
            if ctx.layout is None:
                dH = dH.view(n0, n1, H)
        if ctx.layout is not None:
            L, B, Bpad = ctx.layout
            dH = dH.view(L, Bpad, H)[:, :B].transpose(0, 1)
        return (dH,)


def _vocab_stage2(P, PT, h_bf, ldh, w_bf, T, H, V, Hp, Tp, Vp, dw_sink=None, live=None):
    dev = w_bf.device
    wT = transpose_to_bf16(w_bf, V, H, Hp, Vp, out=workspace("vocab_WT", H * Vp, BF16, dev).view(H, Vp))
    with timed("vocab_dH"):
        # rows past the live count are never read (scattered as 0)
        dH = gemm_nt(P, Vp, wT, Vp, T, H, Vp) if live is None else gemm_nt_live_rows(P, Vp, wT, Vp, T, H, Vp, live)
    hT = transpose_to_bf16(h_bf, T, H, ldh, Tp, out=workspace("vocab_hT", H * Tp, BF16, dev).view(H, Tp))
    with timed("vocab_dW"):
        dW = gemm_nt_wgrad(PT, Tp, hT, Tp, V, H, Tp, dw_sink, k_live=live)   # K = token rows: only the live ones
    return dH, dW


def vocab_argmax(hid, weight, bias, inv_temp=1.0, mask_col=-1):
    """Greedy symbol and its log-probability per row of hid [T,H] (decoders.py:252-254 topk(1) on the log-softmax of the
    projection, :247-248 mask_unk) without a [T,V] tensor.  -> (ids int64 [T], logp fp32 [T]); no gradient."""
    with torch.no_grad():
        _need_cuda(hid, weight, bias)
        T, H, V, Hp, h_bf, ldh, w_bf = _vocab_operands(hid, weight)
        dev = weight.device
        ws = workspace("vocab_argmax", _lib.load().pv_vocab_argmax_ws_bytes(T, V), torch.uint8, dev)
        ids = torch.empty(T, dtype=torch.int64, device=dev)
        logp = torch.empty(T, dtype=torch.float32, device=dev)
        _lib.call("pv_vocab_argmax", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias.contiguous()), T, V, Hp, inv_temp, int(mask_col),
                  _p(ws), _p(ids), _p(logp), _stream())
        return ids, logp


VOCAB_TOPK_MAX = 16


def vocab_topk(hid, weight, bias, k, inv_temp=1.0, mask_col=-1):
    """The k best symbols per row of hid [T,H] and their log-probabilities (the rows beam search ranks, decoders.py:263-270)
    without a [T,V] tensor.  -> (ids int64 [T,k], logp fp32 [T,k], descending); no gradient."""
    with torch.no_grad():
        _need_cuda(hid, weight, bias)
        T, H, V, Hp, h_bf, ldh, w_bf = _vocab_operands(hid, weight)
        dev = weight.device
        ws = workspace("vocab_argmax", _lib.load().pv_vocab_argmax_ws_bytes(T, V), torch.uint8, dev)
        ids = torch.empty((T, k), dtype=torch.int64, device=dev)
        logp = torch.empty((T, k), dtype=torch.float32, device=dev)
        _lib.call("pv_vocab_topk", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias.contiguous()), T, V, Hp, inv_temp, int(mask_col), int(k),
                  _p(ws), _p(ids), _p(logp), _stream())
        return ids, logp


def vocab_sample(hid, weight, bias, inv_temp=1.0, mask_col=-1, noise=None, site=0):
    """One multinomial draw per row of hid [T,H] from softmax((hid W^T + b) * inv_temp) (decoders.py:255-256 ->
    base_modules.py:54-59) without a [T,V] tensor: arg-max of (scaled logit - log q), q ~ Exp(1).  noise [T,V]: the q draws
    (parity tests); None: Philox from ops.rng_state (advance it with ops.rng_advance between decodes) and `site`.
    -> (ids int64 [T], logp fp32 [T] = log-probability of the drawn symbol); no gradient."""
    with torch.no_grad():
        _need_cuda(hid, weight, bias)
        T, H, V, Hp, h_bf, ldh, w_bf = _vocab_operands(hid, weight)
        dev = weight.device
        ws = workspace("vocab_sample", _lib.load().pv_vocab_sample_ws_bytes(T, V), torch.uint8, dev)
        ids = torch.empty(T, dtype=torch.int64, device=dev)
        logp = torch.empty(T, dtype=torch.float32, device=dev)
        nq = None if noise is None else noise.to(torch.float32).contiguous()
        _lib.call("pv_vocab_sample", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias.contiguous()), T, V, Hp, inv_temp, int(mask_col),
                  _p(nq), 0 if nq is None else nq.stride(0), _p(rng_state(dev)), int(site), _p(ws), _p(ids), _p(logp),
                  _stream())
        return ids, logp


LAST_DECODE_WS = []     # workspace of the most recent pv_greedy_decode launch, for greedy_decode_check()


def greedy_decode(emb_weight, w_ih, w_hh, b_ih, b_hh, proj_weight, proj_bias, h0, L, sos_id, latent=None, inv_temp=1.0,
                  mask_col=-1, sample=False, site0=0):
    """All L tokens of a free-running greedy decode in one persistent launch (pv_greedy_decode; decoders.py:236-262 for a
    one-layer GRU decoder without attention).  The input projection of every vocabulary entry is one GEMM per call
    (gx_table = emb W_ih[:, :E]^T + b_ih), a latent input adds a per-row constant.  h0 [B,H] ->
    (seq int64 [B,L], logp fp32 [B,L], h_n fp32 [B,H]); no gradient.
    sample=True: gen_type 'sample' in the same launch (pv_sample_decode): token t draws with the device generator
    (ops.rng_state; advance it between decodes) at call site site0 + t -- the draws of ops.vocab_sample(site=site0 + t)."""
    with torch.no_grad():
        _need_cuda(emb_weight, w_ih, w_hh, b_ih, b_hh, proj_weight, proj_bias, h0)
        B, H = h0.shape
        V, E = emb_weight.shape
        dev = h0.device
        Hk = round_up(H, 32)
        Ep = round_up(E, 64)
        emb_bf = pack_matrix(emb_weight, Ep)
        if latent is None:
            assert w_ih.shape[1] == E
            wie_bf = pack_matrix(w_ih, Ep)
        else:
            Z = latent.shape[1]
            assert w_ih.shape[1] == E + Z
            wie_bf = pack_matrix(w_ih[:, :E].contiguous(), Ep)
        table = gemm_nt(emb_bf, Ep, wie_bf, Ep, V, 3 * H, Ep, bias=b_ih.contiguous(),
                        out=workspace("decode_gx_table", V * 3 * H, torch.float32, dev).view(V, 3 * H))
        const = None
        if latent is not None:
            Zp = round_up(Z, 64)
            lat_bf = cast_pack(latent.contiguous(), 1, B, B, Z, 0, Z, 1, Zp)
            const = gemm_nt(lat_bf, Zp, pack_matrix(w_ih[:, E:].contiguous(), Zp), Zp, B, 3 * H, Zp)
        whh_bf = pack_matrix(w_hh, Hk)
        Hp = round_up(H, 64)
        wp_bf = pack_matrix(proj_weight, Hp)
        nbytes = (_lib.load().pv_sample_decode_ws_bytes if sample else _lib.load().pv_greedy_decode_ws_bytes)(B, H, V, L)
        ws = workspace("greedy_decode", nbytes, torch.uint8, dev)
        seq = torch.empty((B, L), dtype=torch.int64, device=dev)
        logp = torch.empty((B, L), dtype=torch.float32, device=dev)
        hn = torch.empty((B, H), dtype=torch.float32, device=dev)
        join_side_work()                                   # the grid needs every CU
        if sample:
            _lib.call("pv_sample_decode", _p(table), _p(const), _p(whh_bf), _p(b_hh.contiguous()), _p(wp_bf), Hp,
                      _p(proj_bias.contiguous()), _p(h0.contiguous()), B, H, V, L, int(sos_id), int(mask_col), float(inv_temp),
                      _p(rng_state(dev)), int(site0), _p(ws), _p(seq), _p(logp), _p(hn), _stream())
        else:
            _lib.call("pv_greedy_decode", _p(table), _p(const), _p(whh_bf), _p(b_hh.contiguous()), _p(wp_bf), Hp,
                      _p(proj_bias.contiguous()), _p(h0.contiguous()), B, H, V, L, int(sos_id), int(mask_col), float(inv_temp),
                      _p(ws), _p(seq), _p(logp), _p(hn), _stream())
        del LAST_DECODE_WS[:]
        LAST_DECODE_WS.append(ws)
        return seq, logp, hn


def greedy_decode_check():
    """Raise if the last pv_greedy_decode launch reported a timed-out inter-workgroup wait (synchronises the device)."""
    while LAST_DECODE_WS:
        ws = LAST_DECODE_WS.pop()
        flag = C.c_int(0)
        _lib.call("pv_greedy_decode_error", _p(ws), C.byref(flag))
        if flag.value:
            raise _lib.PevaeHipError("pv_greedy_decode: a bounded inter-workgroup wait timed out")


class VocabLogProbDense(torch.autograd.Function):
    """Dense [T,V] log-softmax rows: the reference's DecoderRNN return value (decoders.py:193) for callers
    that want it as a tensor.  Costs the 4 B x T x V write the fused path avoids."""

    @staticmethod
    def forward(ctx, hid, weight, bias, inv_temp):
        _need_cuda(hid, weight, bias)
        T, H, V, Hp, h_bf, ldh, w_bf = _vocab_operands(hid, weight)
        dev = weight.device
        ws = workspace("vocab_fwd", _lib.load().pv_vocab_nll_fwd_ws_bytes(T, V), torch.uint8, dev)
        lse = torch.empty(T, dtype=torch.float32, device=dev)
        nll = workspace("vocab_nll_scratch", T, torch.float32, dev)
        tgt = torch.zeros(T, dtype=torch.int64, device=dev)
        bias_c = bias.contiguous()
        _lib.call("pv_vocab_nll_fwd", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias_c), _p(tgt), T, V, Hp, inv_temp, -1,
                  _p(ws), _p(lse), _p(nll), None, _stream())
        logp = torch.empty((T, V), dtype=torch.float32, device=dev)
        _lib.call("pv_vocab_logprob_dense", _p(h_bf), ldh, _p(w_bf), Hp, _p(bias_c), _p(lse), T, V, Hp, inv_temp,
                  _p(logp), _stream())
        ctx.save_for_backward(h_bf, w_bf, logp)
        ctx.dims = (T, H, V, Hp, ldh, inv_temp)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        h_bf, w_bf, logp = ctx.saved_tensors
        T, H, V, Hp, ldh, inv_temp = ctx.dims
        dev = w_bf.device
        Tp, Vp = round_up(T, 64), round_up(V, 64)
        P = workspace("vocab_P", Tp * Vp, BF16, dev)
        dbias = torch.empty(V, dtype=torch.float32, device=dev)
        _lib.call("pv_logsoftmax_bwd_bf16", _p(logp), _p(dlogp.contiguous()), T, V, inv_temp, _p(P), Vp, _p(dbias),
                  _stream())
        PT = transpose_to_bf16(P, T, V, Vp, Tp, out=workspace("vocab_PT", Vp * Tp, BF16, dev)[:V * Tp].view(V, Tp))
        dH, dW = _vocab_stage2(P, PT, h_bf, ldh, w_bf, T, H, V, Hp, Tp, Vp)
        return dH, dW, dbias, None


# ------------------------------------------------------------------------------------------------------------
# GRU (input projection GEMM + persistent recurrence)
# ------------------------------------------------------------------------------------------------------------

GRU_MAX_WGS = 0     # 0 = one workgroup per CU of the current device
GRU_FRAGMENT_EXCHANGE = os.environ.get("PEVAE_GRU_DGX", "1") != "0"
import collections

# (sync buffer, NG, tag) of the most recent launches, for gru_check_all() (tests, smoke, bench, engine evaluations):
# a bounded ring -- the oldest entry is dropped, so later launches stay observable and at most 64 sync buffers are held
LAST_GRU_SYNC = collections.deque(maxlen=64)


def gru_check_all():
    """Raise if any recent persistent-GRU launch, or a Block forward's cluster hand-off, reported a timed-out wait
    (synchronises the device)."""
    while LAST_GRU_SYNC:
        sync, NG, tag = LAST_GRU_SYNC.pop()
        gru_check_sync(sync, NG, tag)
    rnn_check()
    block_check()
    greedy_decode_check()


def rnn_check():
    """Raise if ANY persistent GRU / LSTM launch since the last check reported a timed-out wait -- the library's sticky
    word, which launches replayed from a HIP graph cannot erase (their own error words are zeroed by every replay, and
    no Python runs during a replay to register them in LAST_GRU_SYNC).  Synchronises the device; clears the word."""
    flag = C.c_int(0)
    _lib.call("pv_rnn_sync_error", C.byref(flag), 1)
    if flag.value:
        raise _lib.PevaeHipError("pv_gru / pv_lstm: a bounded inter-workgroup wait timed out in a launch since the last "
                                 "check (possibly inside a replayed HIP graph): the recurrent states are garbage")


def block_check():
    """Raise if an inter-workgroup wait of pv_block_fwd has given up (synchronises the device)."""
    flag = C.c_int(0)
    _lib.call("pv_block_sync_error", C.byref(flag))
    if flag.value:
        raise _lib.PevaeHipError("pv_block_fwd: a bounded inter-workgroup wait timed out")


def block_rows(rows=-1, wide=-1):
    """Select the fused Block kernels: rows 4 (default: 4-row slabs) or 16 (the round-2 16-row slabs / clusters); wide: 4-row
    kernels also for launches with a 512-wide Block.  -> rows now in force.  Not between a forward pass and its backward."""
    r = _lib.load().pv_block_rows(int(rows), int(wide))
    if r < 0:
        raise _lib.PevaeHipError("pv_block_rows: rows must be 4 or 16")
    return r


def gru_check_sync(sync, NG, what):
    """Raise if the persistent kernel reported a timed-out spin (debug / tests; forces a device sync)."""
    flag = int(sync.view(torch.int32)[NG].item())
    if flag != 0:
        raise _lib.PevaeHipError("%s: a bounded inter-workgroup wait timed out (err=%d)" % (what, flag))


class GRULayer(torch.autograd.Function):
    """One unidirectional GRU layer over x [B,L,E] (batch_first), torch.nn.GRU semantics
    (base_modules.py:42; encoders.py:59-62; decoders.py:182).  Returns (out [B,L,H], h_n [B,H],
    hseq bf16 [(L+1)*Bpad, Hk] = every state in time-major order, reusable as a GEMM operand)."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh, save, lengths=None, x_side=False, dx_consumer=None):
        _need_cuda(x, w_ih, w_hh, b_ih, b_hh, h0, lengths)
        B, L, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        ctx.x_side = bool(x_side)           # x was produced under on_side(): its gradient's consumers run on the side stream
        ctx.dx_consumer = dx_consumer       # x is a leaf value (embed_dropout_leaf): this closure takes dL/dx
        Hk, K3, Bpad, NG = gru_geometry(B, H)
        Ep = round_up(E, 64)
        pre = _PREPROJ.pop((x.data_ptr(), tuple(x.shape), w_ih.data_ptr()), None) if _PREPROJ else None
        whh_bf = pack_matrix(w_hh, Hk)
        if pre is not None:                 # made ahead on the side stream (preproject_gru_input): joined below
            x_bf, xprojT = pre
        else:
            x_bf = cast_pack(x, L, B, Bpad, E, x.stride(1), x.stride(0), x.stride(2), Ep)
            wih_bf = pack_matrix(w_ih, Ep)
            # xproj^T [3H][L*Bpad]: the weight is the A operand, so a lane of the recurrence finds its 4 batch rows in 16 bytes
            xprojT = gemm_nt_ct(x_bf, Ep, wih_bf, Ep, L * Bpad, 3 * H, Ep, b_ih.contiguous(),
                                workspace("gru_xprojT", 3 * H * L * Bpad, torch.float32, dev).view(3 * H, L * Bpad))
        hseq = torch.empty(((L + 1) * Bpad, Hk), dtype=BF16, device=dev)
        hn = torch.empty((B, H), dtype=torch.float32, device=dev)
        gsave = torch.empty((L * 4 * H, Bpad), dtype=torch.float32, device=dev) if save else None
        hall = torch.empty(((L + 1), Bpad, H), dtype=torch.float32, device=dev)
        sync = torch.empty(_lib.load().pv_gru_sync_bytes(B), dtype=torch.uint8, device=dev)
        h0c = None if h0 is None else h0.contiguous()
        if not (FWD_RNN_BESIDE_SIDE and pre is None and h0 is None):
            join_side_work()
        # (else: the encoder's launch -- no initial state, nothing prepared ahead -- does not wait for the side stream: the
        # forward recurrence holds 128 of the 256 CUs, so the head's side work runs beside it; its consumers join later)
        with timed("gru_fwd"):
            _lib.call("pv_gru_fwd", _p(xprojT), L * Bpad, _p(whh_bf), _p(b_hh.contiguous()), _p(h0c), _p(lengths), L, B, H, 0,
                      _p(hseq), _p(gsave), _p(hall), _p(hn), _p(sync), GRU_MAX_WGS, _stream())
        out = hall[1:, :B].transpose(0, 1)          # [B,L,H] view of the fp32 states: no second copy is written
        if save:
            ctx.save_for_backward(x_bf, hseq, gsave, hall, w_ih, w_hh)
        LAST_GRU_SYNC.append((sync, NG, "pv_gru_fwd"))
        ctx.dims = (B, L, E, H, Hk, K3, Bpad, NG, Ep)
        ctx.params = (w_ih, w_hh)
        ctx.bias_params = (b_ih, b_hh)
        ctx.has_h0 = h0 is not None
        ctx.sync = sync
        ctx.mark_non_differentiable(hseq)
        ctx.set_materialize_grads(False)    # an unused `out` / `h_n` (and hseq) arrives as None, not as a zero-filled tensor
        return out, hn, hseq

    @staticmethod
    def backward(ctx, dout, dhn, _dhseq):
        x_bf, hseq, gsave, hall, w_ih, w_hh = ctx.saved_tensors
        B, L, E, H, Hk, K3, Bpad, NG, Ep = ctx.dims
        dev = x_bf.device
        whhT = transpose_to_bf16(w_hh.contiguous(), 3 * H, H, H, K3)
        dgi = torch.empty((L * Bpad, K3), dtype=BF16, device=dev)
        dgh = torch.empty((L * Bpad, K3), dtype=BF16, device=dev)
        # the copy of dgh the workgroups exchange, in MFMA-fragment order (scratch of the kernel; PEVAE_GRU_DGX=0: none)
        dgx = workspace("gru_dgx", L * Bpad * K3, BF16, dev) if GRU_FRAGMENT_EXCHANGE else None
        dh0 = torch.empty((B, H), dtype=torch.float32, device=dev)
        dbp = workspace("gru_dbias_part", 2 * NG * 4 * H, torch.float32, dev)
        db_ih = torch.empty(3 * H, dtype=torch.float32, device=dev)
        db_hh = torch.empty(3 * H, dtype=torch.float32, device=dev)
        sync = torch.empty(_lib.load().pv_gru_sync_bytes(B), dtype=torch.uint8, device=dev)
        if dout is not None:
            dout = dout if dout.stride(2) == 1 else dout.contiguous()
        dhn_c = None if dhn is None else dhn.contiguous()
        join_side_work()        # the recurrence's workgroups wait for each other: nothing else may hold CUs ...
        beside = launch_beside_recurrence()      # ... but a long full-chip kernel that asked for it: half the chip each
        with timed("gru_bwd"):
            _lib.call("pv_gru_bwd", _p(dout), 0 if dout is None else dout.stride(0),
                      0 if dout is None else dout.stride(1), _p(dhn_c), _p(gsave), _p(hall), _p(whhT), L, B, H, 0,
                      _p(dgi), _p(dgh), _p(dgx), _p(dh0), _p(dbp), _p(db_ih), _p(db_hh), _p(sync),
                      min(beside, GRU_MAX_WGS) if beside and GRU_MAX_WGS else (beside or GRU_MAX_WGS), _stream())
        LAST_GRU_SYNC.append((sync, NG, "pv_gru_bwd"))
        launch_after_recurrence()
        Tt = L * Bpad
        Tp = round_up(Tt, 64)
        p_ih, p_hh = ctx.params
        s_ih, s_hh = grad_sink(p_ih, (3 * H, E)), grad_sink(p_hh, (3 * H, H))
        sinks = s_ih is not None and s_hh is not None
        # dx = dgi W_ih  (time-major rows), then back to [B,L,E]; off the critical path when its consumers are (x_side),
        # or handed to the closure of a leaf input inside a postponed launch (dx_consumer)
        def input_grad():
            wihT = transpose_to_bf16(w_ih.contiguous(), 3 * H, E, E, K3)
            dx_tm = gemm_nt(dgi, K3, wihT, K3, Tt, E, K3)
            return dx_tm.view(L, Bpad, E)[:, :B].transpose(0, 1)

        if ctx.dx_consumer is not None:
            consumer = ctx.dx_consumer
            defer_side_work(lambda: consumer(input_grad()), keep=(dgi,))
            dx = None
        else:
            with side_work(sinks and ctx.x_side and SIDE_DX, keep=(dgi,)):
                dx = input_grad()
        # dW_ih = dgi^T x ; dW_hh = dgh^T h_prev   (K = all time-major rows).  Nothing downstream reads them before the
        # optimizer: with gradient sinks they run beside whatever the backward pass does next (the latent stack's chain of
        # small kernels, the embedding gradients)
        def wgrads():
            if GEMM_TN and Tt % 64 == 0:
                # token-major operands as they are: dgi / dgh [Tt][K3], x_bf [Tt][Ep], h_prev = the first Tt rows of hseq
                r_ih = gemm_tn_wgrad(dgi, K3, x_bf, Ep, 3 * H, E, Tt, s_ih, ws_tag="gemm_splitk_rnn")
                r_hh = gemm_tn_wgrad(dgh, K3, hseq, Hk, 3 * H, H, Tt, s_hh, ws_tag="gemm_splitk_rnn")
                return r_ih, r_hh
            dgiT = transpose_to_bf16(dgi, Tt, 3 * H, K3, Tp, out=workspace("gru_dgT", 3 * H * Tp, BF16, dev).view(3 * H, Tp))
            xT = transpose_to_bf16(x_bf, Tt, E, Ep, Tp, out=workspace("gru_xT", E * Tp, BF16, dev).view(E, Tp))
            r_ih = gemm_nt_wgrad(dgiT, Tp, xT, Tp, 3 * H, E, Tp, s_ih, ws_tag="gemm_splitk_rnn")
            dghT = transpose_to_bf16(dgh, Tt, 3 * H, K3, Tp, out=workspace("gru_dgT", 3 * H * Tp, BF16, dev).view(3 * H, Tp))
            hpT = transpose_to_bf16(hseq, Tt, H, Hk, Tp, out=workspace("gru_hpT", H * Tp, BF16, dev).view(H, Tp))
            r_hh = gemm_nt_wgrad(dghT, Tp, hpT, Tp, 3 * H, H, Tp, s_hh, ws_tag="gemm_splitk_rnn")
            return r_ih, r_hh

        if sinks:
            defer_side_work(wgrads, keep=(dgi, dgh, x_bf, hseq))
            dw_ih = dw_hh = None
        else:
            dw_ih, dw_hh = wgrads()
        # the bias gradients: with sinks, the two accumulation adds leave the critical path as well (autograd would run
        # them as AccumulateGrad kernels right behind this node)
        sb_ih, sb_hh = ctx.bias_params
        k_ih, k_hh = grad_sink(sb_ih, (3 * H,)), grad_sink(sb_hh, (3 * H,))
        if sinks and k_ih is not None and k_hh is not None and ASYNC_WGRAD:
            def bias_adds(a=db_ih, b=db_hh):
                k_ih.add_(a)
                k_hh.add_(b)
            defer_side_work(bias_adds, keep=(db_ih, db_hh))
            db_ih = db_hh = None
        return dx, (dh0 if ctx.has_h0 else None), dw_ih, dw_hh, db_ih, db_hh, None, None, None, None


class LSTMLayer(torch.autograd.Function):
    """One unidirectional LSTM layer over x [B,L,E] (batch_first), torch.nn.LSTM semantics (rnn_cell 'lstm':
    base_modules.py:42-51; encoders.py:59-62; decoders.py:182).  Returns (out [B,L,H], h_n [B,H], c_n [B,H],
    hseq bf16 as GRULayer).  One gate-gradient image feeds dx, dW_ih and dW_hh; db_ih == db_hh."""

    @staticmethod
    def forward(ctx, x, h0, c0, w_ih, w_hh, b_ih, b_hh, save, lengths=None):
        _need_cuda(x, w_ih, w_hh, b_ih, b_hh, h0, c0, lengths)
        B, L, E = x.shape
        H = w_hh.shape[1]
        dev = x.device
        Hk, K4, Bpad, NG = lstm_geometry(B, H)
        Ep = round_up(E, 64)
        x_bf = cast_pack(x, L, B, Bpad, E, x.stride(1), x.stride(0), x.stride(2), Ep)
        wih_bf = pack_matrix(w_ih, Ep)
        whh_bf = pack_matrix(w_hh, Hk)
        xprojT = gemm_nt_ct(x_bf, Ep, wih_bf, Ep, L * Bpad, 4 * H, Ep, b_ih.contiguous(),
                            workspace("lstm_xprojT", 4 * H * L * Bpad, torch.float32, dev).view(4 * H, L * Bpad))
        hseq = torch.empty(((L + 1) * Bpad, Hk), dtype=BF16, device=dev)
        hn = torch.empty((B, H), dtype=torch.float32, device=dev)
        cn = torch.empty((B, H), dtype=torch.float32, device=dev)
        gsave = torch.empty((L * 4 * H, Bpad), dtype=torch.float32, device=dev) if save else None
        hall = torch.empty(((L + 1), Bpad, H), dtype=torch.float32, device=dev)
        call = torch.empty(((L + 1), Bpad, H), dtype=torch.float32, device=dev)
        sync = torch.empty(_lib.load().pv_gru_sync_bytes(B), dtype=torch.uint8, device=dev)
        h0c = None if h0 is None else h0.contiguous()
        c0c = None if c0 is None else c0.contiguous()
        with timed("lstm_fwd"):
            _lib.call("pv_lstm_fwd", _p(xprojT), L * Bpad, _p(whh_bf), _p(b_hh.contiguous()), _p(h0c), _p(c0c), _p(lengths),
                      L, B, H, 0,
                      _p(hseq), _p(gsave), _p(hall), _p(call), _p(hn), _p(cn), _p(sync), GRU_MAX_WGS, _stream())
        out = hall[1:, :B].transpose(0, 1)
        if save:
            ctx.save_for_backward(x_bf, hseq, gsave, hall, call, w_ih, w_hh)
        LAST_GRU_SYNC.append((sync, NG, "pv_lstm_fwd"))
        ctx.dims = (B, L, E, H, Hk, K4, Bpad, NG, Ep)
        ctx.params = (w_ih, w_hh)
        ctx.lengths = lengths
        ctx.has_state = (h0 is not None, c0 is not None)
        ctx.mark_non_differentiable(hseq)
        ctx.set_materialize_grads(False)
        return out, hn, cn, hseq

    @staticmethod
    def backward(ctx, dout, dhn, dcn, _dhseq):
        x_bf, hseq, gsave, hall, call, w_ih, w_hh = ctx.saved_tensors
        B, L, E, H, Hk, K4, Bpad, NG, Ep = ctx.dims
        dev = x_bf.device
        whhT = transpose_to_bf16(w_hh.contiguous(), 4 * H, H, H, K4)
        dg = torch.empty((L * Bpad, K4), dtype=BF16, device=dev)
        dh0 = torch.empty((B, H), dtype=torch.float32, device=dev)
        dc0 = torch.empty((B, H), dtype=torch.float32, device=dev)
        dbp = workspace("gru_dbias_part", 2 * NG * 4 * H, torch.float32, dev)
        db_ih = torch.empty(4 * H, dtype=torch.float32, device=dev)
        db_hh = torch.empty(4 * H, dtype=torch.float32, device=dev)
        sync = torch.empty(_lib.load().pv_gru_sync_bytes(B), dtype=torch.uint8, device=dev)
        if dout is not None:
            dout = dout if dout.stride(2) == 1 else dout.contiguous()
        dhn_c = None if dhn is None else dhn.contiguous()
        dcn_c = None if dcn is None else dcn.contiguous()
        with timed("lstm_bwd"):
            _lib.call("pv_lstm_bwd", _p(dout), 0 if dout is None else dout.stride(0),
                      0 if dout is None else dout.stride(1), _p(dhn_c), _p(dcn_c), _p(gsave), _p(hall), _p(call),
                      _p(whhT), L, B, H, 0, _p(dg), _p(dh0), _p(dc0), _p(dbp), _p(db_ih), _p(db_hh), _p(ctx.lengths),
                      _p(sync), GRU_MAX_WGS, _stream())
        LAST_GRU_SYNC.append((sync, NG, "pv_lstm_bwd"))
        Tt = L * Bpad
        Tp = round_up(Tt, 64)
        wihT = transpose_to_bf16(w_ih.contiguous(), 4 * H, E, E, K4)
        dx_tm = gemm_nt(dg, K4, wihT, K4, Tt, E, K4)
        dx = dx_tm.view(L, Bpad, E)[:, :B].transpose(0, 1)
        p_ih, p_hh = ctx.params
        if GEMM_TN and Tt % 64 == 0:
            dw_ih = gemm_tn_wgrad(dg, K4, x_bf, Ep, 4 * H, E, Tt, grad_sink(p_ih, (4 * H, E)))
            dw_hh = gemm_tn_wgrad(dg, K4, hseq, Hk, 4 * H, H, Tt, grad_sink(p_hh, (4 * H, H)))
        else:
            dgT = transpose_to_bf16(dg, Tt, 4 * H, K4, Tp, out=workspace("lstm_dgT", 4 * H * Tp, BF16, dev).view(4 * H, Tp))
            xT = transpose_to_bf16(x_bf, Tt, E, Ep, Tp, out=workspace("gru_xT", E * Tp, BF16, dev).view(E, Tp))
            hpT = transpose_to_bf16(hseq, Tt, H, Hk, Tp, out=workspace("gru_hpT", H * Tp, BF16, dev).view(H, Tp))
            dw_ih = gemm_nt_wgrad(dgT, Tp, xT, Tp, 4 * H, E, Tp, grad_sink(p_ih, (4 * H, E)))
            dw_hh = gemm_nt_wgrad(dgT, Tp, hpT, Tp, 4 * H, H, Tp, grad_sink(p_hh, (4 * H, H)))
        has_h0, has_c0 = ctx.has_state
        return dx, (dh0 if has_h0 else None), (dc0 if has_c0 else None), dw_ih, dw_hh, db_ih, db_hh, None, None


# ------------------------------------------------------------------------------------------------------------
# embedding
# ------------------------------------------------------------------------------------------------------------

# The ids of a lookup are known when the step starts, the gradient of its rows only at the tail of the backward pass (the
# encoder-side tables come last).  With a gradient sink and PRESORT on, the stable sort of the (row id, occurrence) pairs
# runs at FORWARD time on a background stream that only the lookup's own backward waits for -- beside the encoder
# recurrence, which leaves half the chip idle -- and the backward pass is left with the segmented sum.  PRESORT_OK is
# cleared by callers whose forward and backward are captured into different HIP graphs (a fork must join inside its graph).
PRESORT = os.environ.get("PEVAE_EMB_PRESORT", "0") == "1"      # measured at the Yelp shape: 2.963 vs 2.891 ms per step WITH it (five more cross-queue edges in the replayed graph cost more than the sorts at the tail): off
PRESORT_OK = True
_BG = {}


def _bg_stream(dev):
    st = _BG.get(dev)
    if st is None:
        st = _BG[dev] = torch.cuda.Stream(device=dev)
    return st


class EmbeddingLookup(torch.autograd.Function):
    """nn.Embedding forward/backward (dialog_models.py:758-765, decoders.py:174): coalesced row gather;
    backward = stable sort + segmented sum (no atomics, bitwise reproducible); padding row gets zero grad."""

    @staticmethod
    def forward(ctx, weight, ids, padding_idx):
        _need_cuda(weight, ids)
        ids_c = ids.contiguous()
        n, D = ids_c.numel(), weight.shape[1]
        w = weight.contiguous()
        out = torch.empty(tuple(ids.shape) + (D,), dtype=torch.float32, device=weight.device)
        if n:
            _lib.call("pv_embedding_gather", _p(w), w.shape[0], D, _p(ids_c), n, n, n, 0, 1, _p(out), D, None, 0,
                      None, _stream())
        ctx.presorted = None
        if (PRESORT and PRESORT_OK and n and ctx.needs_input_grad[0]       # (grad mode is off inside a Function's forward)
                and grad_sink(weight, tuple(weight.shape)) is not None):
            dev = weight.device
            nbytes = _lib.load().pv_embedding_bwd_ws_bytes(n, D)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)         # this lookup's own: it lives until its backward
            cur, bg = torch.cuda.current_stream(dev), _bg_stream(dev)
            bg.wait_stream(cur)
            with torch.cuda.stream(bg):
                _lib.call("pv_embedding_sort", _p(ids_c), n, n, n, 0, 1, w.shape[0], D, _p(ws), nbytes, _stream())
                ev = torch.cuda.Event()
                ev.record(bg)
            ctx.presorted = (ws, nbytes, ev)
        ctx.save_for_backward(ids_c)
        ctx.meta = (w.shape[0], D, -1 if padding_idx is None else padding_idx)
        ctx.param = weight
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids_c,) = ctx.saved_tensors
        rows, D, pad = ctx.meta
        sink = grad_sink(ctx.param, (rows, D))
        dw = embedding_backward(dout, ids_c, rows, D, pad, sink, ctx.presorted if sink is not None else None)
        ctx.presorted = None
        return (None if sink is not None else dw), None, None


def embedding_backward(dout, ids_c, rows, D, pad, sink, presorted=None):
    """Table gradient of a lookup: stable sort + segmented sum, accumulated into `sink` when given (else returned).
    presorted = (workspace, bytes, event) of a pv_embedding_sort made at forward time: only the segmented sum is left."""
    n = ids_c.numel()
    dev = dout.device
    dw = sink if sink is not None else torch.empty((rows, D), dtype=torch.float32, device=dev)
    if dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= D and dout.stride(0) % 4 == 0:
        g, ldg = dout, dout.stride(0)         # a column slice of a wider gradient (torch.cat backward): no copy
    else:
        g, ldg = dout.contiguous().view(-1, D), D
    if presorted is None and _PRESORTED and sink is not None:
        hit = _PRESORTED.pop((ids_c.data_ptr(), ids_c._version, n, rows, D), None)
        if hit is not None:                # presort_lookup(): ordered by the forward pass's own join, no event
            _lib.call("pv_embedding_bwd_sorted", _p(g), ldg, n, rows, D, pad, 1, _p(dw), _p(hit[0]), hit[1], _stream())
            return dw
    if presorted is not None:
        ws, nbytes, ev = presorted
        torch.cuda.current_stream(dev).wait_event(ev)
        _lib.call("pv_embedding_bwd_sorted", _p(g), ldg, n, rows, D, pad, 1 if sink is not None else 0, _p(dw), _p(ws), nbytes,
                  _stream())
        return dw
    nbytes = _lib.load().pv_embedding_bwd_ws_bytes(max(n, 1), D)
    ws = workspace("emb_bwd", nbytes, torch.uint8, dev)
    _lib.call("pv_embedding_bwd", _p(g), ldg, _p(ids_c), n, max(n, 1), max(n, 1), 0, 1, rows, D, pad,
              1 if sink is not None else 0, _p(dw), _p(ws), nbytes, _stream())
    return dw


def embed_dropout_leaf(weight, ids, padding_idx, drop):
    """Teacher-forced decoder input (decoders.py:174-175): dropout(embedding(ids)) as ONE value without autograd nodes,
    plus the closure that takes dL/d(that value) to the table's gradient sink.  GRULayer hands the closure its dx inside a
    postponed side-stream launch (defer_side_work): the sort and the segmented sum of the embedding gradient, which
    nothing downstream waits for, then never stand between the recurrence's backward and the latent stack's.
    Returns None when the table has no gradient sink (the ordinary autograd path is used then)."""
    rows, D = weight.shape
    sink = grad_sink(weight, (rows, D))
    if sink is None or not torch.is_grad_enabled():
        return None
    pad = -1 if padding_idx is None else padding_idx
    dropping = drop is not None and drop.training and drop.p > 0
    with torch.no_grad():
        ids_c = ids.contiguous()
        n = ids_c.numel()
        w = weight.contiguous()
        x = torch.empty(tuple(ids.shape) + (D,), dtype=torch.float32, device=weight.device)
        _lib.call("pv_embedding_gather", _p(w), rows, D, _p(ids_c), n, n, n, 0, 1, _p(x), D, None, 0, None, _stream())
        p = site = mask = None
        if dropping:
            p, site, mask = drop.p, drop.next_site(), drop.mask
            if mask is not None:
                mask = mask.to(torch.uint8).contiguous()
            y = torch.empty_like(x)
            _drop_call(x.view(1, -1, D), y.view(1, -1, D), p, site, mask, (0, D))
            x = y
    Bq, Lq = ids.shape

    def consume(dx):                       # dx [B, L, D], any strides
        if dropping:
            d2 = torch.empty((Bq, Lq, D), dtype=torch.float32, device=dx.device)
            _drop_call(dx, d2, p, site, mask, (Lq * D, D))
            dx = d2
        embedding_backward(dx.reshape(-1, D) if dx.is_contiguous() else dx.contiguous().view(-1, D), ids_c, rows, D, pad,
                           grad_sink(weight, (rows, D)))

    return x, consume


def embedding(weight, ids, padding_idx=None):
    return EmbeddingLookup.apply(weight, ids, padding_idx)


# ------------------------------------------------------------------------------------------------------------
# small exact-fp32 linear
# ------------------------------------------------------------------------------------------------------------

def _lin(A, sam, sak, Bm, sbk, sbn, bias, Cm, ldc, M, N, K, accumulate=0, alpha=1.0, batch=1, bsa=0, bsb=0, bsc=0,
         alpha_dev=None):
    _lib.call("pv_linear_f32", _p(A), sam, sak, _p(Bm), sbk, sbn, _p(bias), _p(Cm), ldc, M, N, K, alpha, _p(alpha_dev),
              accumulate, batch, bsa, bsb, bsc, _stream())


# Off: measured 2.670 vs 2.664 ms per step with it (twice each) -- the only generic linear on the step's critical chain is
# prior_fc, whose backward stands behind the join in front of the encoder recurrence either way, and the two extra
# cross-queue edges cost what the 30 us of launches saved.
LINEAR_WGRAD_SIDE = os.environ.get("PEVAE_LINEAR_WGRAD_SIDE", "0") == "1"


class LinearF32(torch.autograd.Function):
    """y = x W^T + b in exact fp32 (Hidden2Gaussian nn_lib.py:198-199, LinearConnector nn_lib.py:143,
    Block linears dialog_models.py:258-263)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_cuda(x, weight, bias)
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K).contiguous()
        M = x2.shape[0]
        w = weight.contiguous()
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _lin(x2, K, 1, w, 1, K, None if bias is None else bias.contiguous(), y, N, M, N, K)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        ctx.params = (weight, bias)
        return y.view(tuple(x.shape[:-1]) + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(M, N).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            _lin(dy2, N, 1, w, K, 1, None, dx, K, M, K, N)                 # dx = dy W
            dx = dx.view(ctx.xshape)
        p_w, p_b = ctx.params
        # With gradient sinks inside DLSVAE.backward() the parameter gradients leave the critical chain (postponed
        # side-stream launches, as the recurrences' and the Blocks'): nothing downstream reads them before the optimizer
        if ctx.needs_input_grad[1]:
            sink = grad_sink(p_w, (N, K))
            if sink is not None and ASYNC_WGRAD and LINEAR_WGRAD_SIDE:
                defer_side_work(lambda s=sink: _lin(dy2, 1, N, x2, K, 1, None, s, K, N, K, M, accumulate=1), keep=(dy2, x2))
            else:
                dw = sink if sink is not None else torch.empty((N, K), dtype=torch.float32, device=dy.device)
                _lin(dy2, 1, N, x2, K, 1, None, dw, K, N, K, M, accumulate=1 if sink is not None else 0)   # dW = dy^T x
                dw = None if sink is not None else dw
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sink = grad_sink(p_b, (N,))
            if sink is not None and ASYNC_WGRAD and LINEAR_WGRAD_SIDE:
                defer_side_work(lambda s=sink: _lib.call("pv_colsum_f32", _p(dy2), M, N, N, _p(s), 1, _stream()), keep=(dy2,))
            else:
                db = sink if sink is not None else torch.empty(N, dtype=torch.float32, device=dy.device)
                _lib.call("pv_colsum_f32", _p(dy2), M, N, N, _p(db), 1 if sink is not None else 0, _stream())
                db = None if sink is not None else db
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearF32.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------------------
# latent connectors
# ------------------------------------------------------------------------------------------------------------

class ReparamKL(torch.autograd.Function):
    """z = mu_q + exp(lv_q) * eps and the row-wise Gaussian KL in one pass (mode 0: NormKLLoss
    criterions.py:145-155; mode 1: gaussian_analytical_kl dialog_models.py:3215-3217)."""

    @staticmethod
    def forward(ctx, mu_q, lv_q, mu_p, lv_p, eps, mode, want_z, want_kl):
        _need_cuda(mu_q, lv_q, mu_p, lv_p, eps)
        B, Z = mu_q.shape
        ts = [None if t is None else t.contiguous() for t in (mu_q, lv_q, mu_p, lv_p, eps)]
        dev = mu_q.device
        z = torch.empty((B, Z), dtype=torch.float32, device=dev) if want_z else None
        kl = torch.empty(B, dtype=torch.float32, device=dev) if want_kl else None
        _lib.call("pv_reparam_kl_fwd", _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), _p(ts[4]), B, Z, mode, _p(z),
                  _p(kl), _stream())
        ctx.ts = ts
        ctx.mode = mode
        outs = (z if want_z else torch.empty(0, device=dev), kl if want_kl else torch.empty(0, device=dev))
        return outs

    @staticmethod
    def backward(ctx, dz, dkl):
        mq, lq, mp, lp, eps = ctx.ts
        B, Z = mq.shape
        dev = mq.device
        dz = None if (dz is None or dz.numel() == 0) else dz.contiguous()
        dkl = None if (dkl is None or dkl.numel() == 0) else dkl.contiguous()
        dmq = torch.empty_like(mq)
        dlq = torch.empty_like(lq)
        dmp = torch.empty_like(mp) if mp is not None else None
        dlp = torch.empty_like(lp) if lp is not None else None
        _lib.call("pv_reparam_kl_bwd", _p(mq), _p(lq), _p(mp), _p(lp), _p(eps), _p(dz), _p(dkl), B, Z, ctx.mode,
                  _p(dmq), _p(dlq), _p(dmp), _p(dlp), _stream())
        return dmq, dlq, dmp, dlp, None, None, None, None


def reparam_kl(mu_q, lv_q, mu_p=None, lv_p=None, eps=None, mode=0, want_z=True, want_kl=True):
    z, kl = ReparamKL.apply(mu_q, lv_q, mu_p, lv_p, eps, mode, want_z, want_kl)
    return (z if want_z else None), (kl if want_kl else None)


class GumbelSoftmax(torch.autograd.Function):
    """nn_lib.py:226-265 with the uniform noise `u` supplied (the reference draws it with CPU torch.rand)."""

    @staticmethod
    def forward(ctx, logits, u, tau, hard):
        _need_cuda(logits, u)
        rows, k = logits.shape
        lg, uu = logits.contiguous(), u.contiguous()
        dev = logits.device
        y_soft = torch.empty((rows, k), dtype=torch.float32, device=dev)
        y = torch.empty((rows, k), dtype=torch.float32, device=dev)
        ids = torch.empty((rows, 1), dtype=torch.int64, device=dev)
        _lib.call("pv_gumbel_softmax_fwd", _p(lg), _p(uu), rows, k, tau, 1 if hard else 0, _p(y_soft), _p(y),
                  _p(ids), _stream())
        ctx.save_for_backward(y_soft)
        ctx.tau = tau
        ctx.mark_non_differentiable(ids)
        return y, ids

    @staticmethod
    def backward(ctx, dy, _dids):
        (y_soft,) = ctx.saved_tensors
        rows, k = y_soft.shape
        dl = torch.empty_like(y_soft)
        _lib.call("pv_softmax_bwd_rows", _p(y_soft), _p(dy.contiguous()), rows, k, 1.0 / ctx.tau, _p(dl), _stream())
        return dl, None, None, None


class CatRows(torch.autograd.Function):
    """Per-row sum_j q (log q - log p) (mode 0, criterions.py:168-169) or -sum_j q log q (mode 1, :206-207)."""

    @staticmethod
    def forward(ctx, log_q, log_p, mode):
        _need_cuda(log_q, log_p)
        k = log_q.shape[-1]
        lq = log_q.reshape(-1, k).contiguous()
        rows = lq.shape[0]
        lp = None
        lp_n = 0
        if mode == 0:
            if log_p.numel() == 1:
                lp, lp_n = log_p.reshape(1).contiguous().to(torch.float32), 1
            else:
                lp = log_p.expand_as(log_q).reshape(-1, k).contiguous()
                lp_n = lp.numel()
        out = torch.empty(rows, dtype=torch.float32, device=log_q.device)
        _lib.call("pv_cat_rows_fwd", _p(lq), _p(lp), lp_n, rows, k, mode, _p(out), _stream())
        ctx.save_for_backward(lq, lp if lp is not None else lq)
        ctx.meta = (mode, lp_n, log_q.shape, None if log_p is None else log_p.shape)
        return out.view(log_q.shape[:-1])

    @staticmethod
    def backward(ctx, drow):
        lq, lp = ctx.saved_tensors
        mode, lp_n, qshape, pshape = ctx.meta
        rows, k = lq.shape
        dlq = torch.empty_like(lq)
        want_p = mode == 0 and ctx.needs_input_grad[1] and lp_n == rows * k
        dlp = torch.empty_like(lq) if want_p else None
        _lib.call("pv_cat_rows_bwd", _p(lq), _p(lp) if mode == 0 else None, lp_n, _p(drow.contiguous().view(-1)), rows,
                  k, mode, _p(dlq), _p(dlp), _stream())
        if want_p:
            dlp = dlp.view(qshape)
            if tuple(pshape) != tuple(qshape):
                dlp = dlp.sum_to_size(pshape)
        return dlq.view(qshape), dlp, None


class BMM(torch.autograd.Function):
    """C[b] = A[b] @ B[b] in exact fp32 for [Bt,M,K] x [Bt,K,N] (any strides): the attention bmm's of
    decoders.py:85-107."""

    @staticmethod
    def forward(ctx, a, b):
        _need_cuda(a, b)
        Bt, M, K = a.shape
        N = b.shape[2]
        c = torch.empty((Bt, M, N), dtype=torch.float32, device=a.device)
        _lin(a, a.stride(1), a.stride(2), b, b.stride(1), b.stride(2), None, c, N, M, N, K, batch=Bt,
             bsa=a.stride(0), bsb=b.stride(0), bsc=M * N)
        ctx.save_for_backward(a, b)
        return c

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        Bt, M, K = a.shape
        N = b.shape[2]
        dc = dc.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:      # dA = dC B^T
            da = torch.empty((Bt, M, K), dtype=torch.float32, device=a.device)
            _lin(dc, N, 1, b, b.stride(2), b.stride(1), None, da, K, M, K, N, batch=Bt, bsa=M * N, bsb=b.stride(0),
                 bsc=M * K)
        if ctx.needs_input_grad[1]:      # dB = A^T dC
            db = torch.empty((Bt, K, N), dtype=torch.float32, device=a.device)
            _lin(a, a.stride(2), a.stride(1), dc, N, 1, None, db, N, K, N, M, batch=Bt, bsa=a.stride(0), bsb=M * N,
                 bsc=K * N)
        return da, db


def bmm(a, b):
    return BMM.apply(a, b)


# ------------------------------------------------------------------------------------------------------------
# decoder attention pieces
# ------------------------------------------------------------------------------------------------------------

class AdditiveScores(torch.autograd.Function):
    """score[b,i,j] = qw . tanh(k[b,j] + q[b,i]) + qb (decoders.py:89-95) without the 4-D intermediate."""

    @staticmethod
    def forward(ctx, q, k, qw, qb):
        _need_cuda(q, k, qw, qb)
        q, k, qw = q.contiguous(), k.contiguous(), qw.contiguous()
        B, Lo, D = q.shape
        Li = k.shape[1]
        score = torch.empty((B, Lo, Li), dtype=torch.float32, device=q.device)
        _lib.call("pv_additive_scores_fwd", _p(q), _p(k), _p(qw), _p(qb), B, Lo, Li, D, _p(score), _stream())
        ctx.save_for_backward(q, k, qw)
        ctx.has_qb = qb is not None
        return score

    @staticmethod
    def backward(ctx, ds):
        q, k, qw = ctx.saved_tensors
        B, Lo, D = q.shape
        Li = k.shape[1]
        ds = ds.contiguous()
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        dqw = torch.empty_like(qw)
        part = workspace("attn_dqw_part", B * Lo * D, torch.float32, q.device)
        _lib.call("pv_additive_scores_bwd", _p(q), _p(k), _p(qw), _p(ds), B, Lo, Li, D, _p(dq), _p(dk), _p(part),
                  _p(dqw), _stream())
        dqb = masked_sum(ds.view(-1))[0:1] if ctx.has_qb else None
        return dq, dk, dqw, dqb


def additive_scores(q, k, qw, qb=None):
    return AdditiveScores.apply(q, k, qw, qb)


class SoftmaxLastDim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        n = x.shape[-1]
        x2 = x.contiguous().view(-1, n)
        y = torch.empty_like(x2)
        _lib.call("pv_softmax_rows", _p(x2), x2.shape[0], n, _p(y), _stream())
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        _lib.call("pv_softmax_rows_bwd", _p(y), _p(dy.contiguous()), y.shape[0], y.shape[1], _p(dx), _stream())
        return dx.view(dy.shape)


def softmax_lastdim(x):
    return SoftmaxLastDim.apply(x)


class _Unary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        _need_cuda(x)
        xc = x.contiguous()
        y = torch.empty_like(xc)
        _lib.call("pv_unary_f32", _p(xc), None, xc.numel(), op, _p(y), _stream())
        ctx.save_for_backward(y if op == 0 else xc)
        ctx.op = op
        return y

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        dx = torch.empty_like(s)
        _lib.call("pv_unary_f32", _p(s), _p(dy.contiguous()), s.numel(), ctx.op + 1, _p(dx), _stream())
        return dx, None


def tanh(x):
    return _Unary.apply(x, 0)


def gelu(x):
    return _Unary.apply(x, 2)


# ------------------------------------------------------------------------------------------------------------
# loss assembly helpers
# ------------------------------------------------------------------------------------------------------------

class MaskedSum(torch.autograd.Function):
    """[sum_i x[i] over mask[i] != ignore, count] as a 2-element device tensor (single fixed-order reduction)."""

    @staticmethod
    def forward(ctx, x, mask, ignore):
        _need_cuda(x, mask)
        xc = x.contiguous().view(-1)
        mc = None if mask is None else mask.contiguous().view(-1)
        out = masked_sum(xc, mc, ignore)
        ctx.save_for_backward(mc if mc is not None else xc)
        ctx.meta = (mask is not None, ignore, x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (m,) = ctx.saved_tensors
        has_mask, ignore, shape = ctx.meta
        dx = torch.empty(m.numel(), dtype=torch.float32, device=m.device)
        _lib.call("pv_masked_sum_bwd", _p(m) if has_mask else None, ignore, _p(dout.contiguous()), m.numel(), _p(dx), _stream())
        return dx.view(shape), None, None


class NllHead(torch.autograd.Function):
    """(NLLEntropy 'seq', Perplexity) of one step from MaskedSum's pair {sum of token NLL, live tokens}: total / batch and
    exp(total / count) (criterions.py:64, :137) in ONE launch, and one launch back -- the stock form is three scalar
    launches forward (div, div, exp) and three backward (div, and the zero-fill + copy of indexing the pair)."""

    @staticmethod
    def forward(ctx, sums, batch):
        _need_cuda(sums)
        out = torch.empty(2, dtype=torch.float32, device=sums.device)
        _lib.call("pv_nll_head_fwd", _p(sums), int(batch), _p(out), _stream())
        ctx.save_for_backward(sums, out)
        ctx.batch = int(batch)
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_nll, g_ppl):
        sums, out = ctx.saved_tensors
        if g_nll is None and g_ppl is None:
            return None, None
        d = torch.empty(2, dtype=torch.float32, device=sums.device)
        _lib.call("pv_nll_head_bwd", _p(sums), _p(out), _p(None if g_nll is None else g_nll.contiguous()),
                  _p(None if g_ppl is None else g_ppl.contiguous()), ctx.batch, _p(d), _stream())
        return d, None


def sum(x):
    return MaskedSum.apply(x, None, 0)[0]


def mean(x):
    return MaskedSum.apply(x, None, 0)[0] / x.numel()


class GatherNLL(torch.autograd.Function):
    """Dense-input path of NLLEntropy/Perplexity (criterions.py:57-63): -logp[t, target[t]], 0 for pads."""

    @staticmethod
    def forward(ctx, logp, target, ignore):
        _need_cuda(logp, target)
        lp = logp.contiguous()
        T, V = lp.shape
        tgt = target.contiguous()
        out = torch.empty(T, dtype=torch.float32, device=lp.device)
        _lib.call("pv_pick_nll_fwd", _p(lp), _p(tgt), ignore, T, V, _p(out), _stream())
        ctx.save_for_backward(tgt)
        ctx.meta = (T, V, ignore)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tgt,) = ctx.saved_tensors
        T, V, ignore = ctx.meta
        dl = torch.empty((T, V), dtype=torch.float32, device=dout.device)
        _lib.call("pv_pick_nll_bwd", _p(dout.contiguous()), _p(tgt), ignore, T, V, _p(dl), _stream())
        return dl, None, None


# ------------------------------------------------------------------------------------------------------------
# DLS latent pyramid pieces (Block / DlsRec / DlsPri / DLS_Auto, dialog_models.py:253-271, :1835-1870, :3161-3352)
# ------------------------------------------------------------------------------------------------------------

class AddLayerNorm(torch.autograd.Function):
    """LayerNorm(x + res) * gamma + beta  (Block: self.layer_norm(x + xhat), dialog_models.py:269)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        _need_cuda(x, res, gamma, beta)
        D = x.shape[-1]
        x2 = x.reshape(-1, D).contiguous()
        r2 = None if res is None else res.reshape(-1, D).contiguous()
        rows = x2.shape[0]
        out = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g, b = gamma.contiguous(), beta.contiguous()
        _lib.call("pv_layernorm_fwd", _p(x2), _p(r2), _p(g), _p(b), rows, D, eps, _p(out), _p(mean), _p(rstd), _stream())
        ctx.save_for_backward(x2, r2 if r2 is not None else x2, g, mean, rstd)
        ctx.has_res = r2 is not None
        ctx.params = (gamma, beta)
        ctx.xshape = x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, r2, g, mean, rstd = ctx.saved_tensors
        rows, D = x2.shape
        d2 = dout.reshape(rows, D).contiguous()
        ds = torch.empty_like(x2)
        p_g, p_b = ctx.params
        sg, sb = grad_sink(p_g, (D,)), grad_sink(p_b, (D,))
        both = sg is not None and sb is not None
        dg = sg if both else torch.empty(D, dtype=torch.float32, device=x2.device)
        db = sb if both else torch.empty(D, dtype=torch.float32, device=x2.device)
        _lib.call("pv_layernorm_bwd", _p(x2), _p(r2) if ctx.has_res else None, _p(g), _p(d2), _p(mean), _p(rstd), rows, D,
                  _p(ds), _p(dg), _p(db), 1 if both else 0, _stream())
        ds = ds.view(ctx.xshape)
        return ds, (ds if ctx.has_res else None), (None if both else dg), (None if both else db), None


def add_layer_norm(x, res, gamma, beta, eps=1e-5):
    return AddLayerNorm.apply(x, res, gamma, beta, eps)


def _lin_fused(A, sam, sak, Bm, sbk, sbn, bias, Cm, ldc, M, N, K, alpha_dev=None, accumulate=0, gelu_out=None,
               gelu_grad=None, r1_c=None, r1_u=None, r1_v=None):
    _lib.call("pv_linear_f32_fused", _p(A), sam, sak, _p(Bm), sbk, sbn, _p(bias), _p(Cm), ldc, M, N, K, 1.0,
              _p(alpha_dev), accumulate, _p(gelu_out), _p(gelu_grad), _p(r1_c), 0 if r1_c is None else r1_c.numel(),
              _p(r1_u), _p(r1_v), _stream())


class SNLinearFn(torch.autograd.Function):
    """y = a (W / sigma)^T + b with sigma = u . (W v) from torch.nn.utils.spectral_norm's power iteration (one per
    TRAINING forward; u / v are constants for autograd).  The GEMM input `a` is either x itself or, for the Block's
    c(gelu(x)) pattern, x_act = gelu(x) that the PRODUCER of x already wrote from its epilogue (`emit_gelu`): every
    GELU is evaluated once per element, and the gradient returned for x is (dy W / sigma) * gelu'(x), applied in the
    dx GEMM's epilogue.  `inv_sigma_pre` carries 1/sigma when the model-wide multi-tensor pre-pass (spectral_prepass)
    has already advanced u / v for this step.  Backward is three launches: one pass over dy (bias gradient +
    <dL/dW_eff, W_eff>), dx, and dW with the rank-1 spectral term in its epilogue, accumulated straight into an existing
    .grad."""

    @staticmethod
    def forward(ctx, x, x_act, weight, bias, u, v, do_iter, eps, emit_gelu, inv_sigma_pre, clone_uv=False):
        _need_cuda(x, x_act, weight, bias, u, v)
        N, K = weight.shape
        x2 = x.reshape(-1, K).contiguous()
        a2 = x2 if x_act is None else x_act.reshape(-1, K).contiguous()
        M = x2.shape[0]
        w = weight.contiguous()
        if inv_sigma_pre is None:
            inv_sigma = torch.empty(1, dtype=torch.float32, device=x.device)
            _lib.call("pv_spectral_norm", _p(w), _p(u), _p(v), N, K, eps, 1 if do_iter else 0, _p(inv_sigma), _stream())
            u, v = u.clone(), v.clone()          # a later forward may advance the buffers before this backward runs
        else:
            inv_sigma = inv_sigma_pre            # u / v advance once per step, before any forward of that step
            if clone_uv:                         # ... unless this module runs again in the same step (its second forward
                u, v = u.clone(), v.clone()      # advances the buffers in place before this call's backward has run)
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gy = torch.empty((M, N), dtype=torch.float32, device=x.device) if emit_gelu else None
        bc = None if bias is None else bias.contiguous()
        _lin_fused(a2, K, 1, w, 1, K, bc, y, N, M, N, K, alpha_dev=inv_sigma, gelu_out=gy)
        ctx.save_for_backward(x2, a2, w, y, bc if bc is not None else inv_sigma, inv_sigma, u, v)
        ctx.set_materialize_grads(False)      # no zero tensor for the non-differentiable gelu output
        ctx.has_bias = bias is not None
        ctx.act_in = x_act is not None
        ctx.params = (weight, bias)
        ctx.xshape = x.shape
        oshape = tuple(x.shape[:-1]) + (N,)
        if emit_gelu:
            gy = gy.view(oshape)
            ctx.mark_non_differentiable(gy)
            return y.view(oshape), gy
        return y.view(oshape), None

    @staticmethod
    def backward(ctx, dy, _dgy):
        x2, a2, w, y, bc, inv_sigma, u, v = ctx.saved_tensors
        if dy is None:
            return (None,) * 11
        M, K = x2.shape
        N = w.shape[0]
        dev = x2.device
        dy2 = dy.reshape(M, N).contiguous()
        p_w, p_b = ctx.params
        dx = dw = db = None
        want_w = ctx.needs_input_grad[2]
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        cpart = torch.empty((N + 63) // 64, dtype=torch.float32, device=dev)
        sink_b = grad_sink(p_b, (N,)) if want_b else None
        if want_b and sink_b is None:
            db = torch.empty(N, dtype=torch.float32, device=dev)
        if want_w or want_b:
            _lib.call("pv_spectral_norm_reduce", _p(dy2), _p(y), _p(bc) if ctx.has_bias else None, M, N,
                      _p(sink_b if sink_b is not None else db), 1 if sink_b is not None else 0, _p(cpart), _stream())
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dev)
            _lin_fused(dy2, N, 1, w, K, 1, None, dx, K, M, K, N, alpha_dev=inv_sigma,
                       gelu_grad=x2 if ctx.act_in else None)                      # dx = (dy W / sigma) * act'(x)
            dx = dx.view(ctx.xshape)
        if want_w:
            # dL/dW_orig = (dy^T a - <dL/dW_eff, W_eff> u v^T) / sigma
            sink = grad_sink(p_w, (N, K))
            dw = sink if sink is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
            _lin_fused(dy2, 1, N, a2, K, 1, None, dw, K, N, K, M, alpha_dev=inv_sigma, accumulate=1 if sink is not None else 0,
                       r1_c=cpart, r1_u=u, r1_v=v)
            dw = None if sink is not None else dw
        return dx, None, dw, db, None, None, None, None, None, None, None


def sn_linear(x, weight_orig, bias, u, v, training, eps=1e-12, x_act=None, emit_gelu=False, inv_sigma=None, clone_uv=False):
    """-> (y, gelu(y) or None)."""
    return SNLinearFn.apply(x, x_act, weight_orig, bias, u, v, bool(training), eps, bool(emit_gelu), inv_sigma, bool(clone_uv))


def spectral_prepass(modules, training, eps=1e-12, cache=None):
    """Advance the power iteration of EVERY spectral-norm weight of a model in one launch (one workgroup per weight)
    and hand each module its 1/sigma for the forward that follows.  `modules`: objects with weight_orig, weight_u,
    weight_v; `cache`: a dict kept by the caller (descriptor table, rebuilt only when a pointer moved)."""
    import numpy as np
    cache = {} if cache is None else cache
    ptrs = tuple((m.weight_orig.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr()) for m in modules)
    if cache.get("ptrs") != ptrs:
        dev = modules[0].weight_orig.device
        inv = torch.empty(len(modules), dtype=torch.float32, device=dev)
        rec = np.zeros((len(modules), 7), dtype=np.int64)
        sizes = [m.weight_orig.numel() for m in modules]
        # W^T of every weight (pv_block_fwd operands), each at a 16-byte aligned offset: tiny weights of the deepest
        # pyramid Blocks (2 x 1, 4 x 2 floats) would otherwise misalign everything behind them
        wt = torch.empty(int(np.sum([(n + 3) // 4 * 4 for n in sizes])), dtype=torch.float32, device=dev)
        wts, off = [], 0
        for i, m in enumerate(modules):
            assert m.weight_orig.is_contiguous()
            n_out, n_in = m.weight_orig.shape
            wts.append(wt[off:off + sizes[i]].view(n_in, n_out))
            rec[i] = (ptrs[i][0], ptrs[i][1], ptrs[i][2], inv.data_ptr() + 4 * i, n_out, n_in, wts[i].data_ptr())
            off += (sizes[i] + 3) // 4 * 4
        max_out, max_in = int(rec[:, 4].max()), int(rec[:, 5].max())
        ws = None
        if max(max_out, max_in) <= 1024:
            ws = torch.empty(_lib.load().pv_spectral_norm_multi_ws_bytes(len(modules), max_out), dtype=torch.uint8, device=dev)
        cache.update(ptrs=ptrs, inv=inv, wts=wts, desc=torch.from_numpy(rec).to(dev), max_out=max_out, max_in=max_in, ws=ws)
    _lib.call("pv_spectral_norm_multi", _p(cache["desc"]), len(modules), cache["max_out"], cache["max_in"], eps,
              1 if training else 0, 1 if FUSED_BLOCK else 0, _p(cache["ws"]), _stream())
    for i, m in enumerate(modules):
        m._inv_sigma_pre = cache["inv"][i:i + 1]
        m._wt_pre = cache["wts"][i]
    return cache


FUSED_BLOCK = True      # Block.forward uses the one-launch kernels below whenever its widths allow it


def block_fusable(din, mid, dout):
    return FUSED_BLOCK and all(d % 16 == 0 and 0 < d <= 512 for d in (din, mid, dout))


def _ptr_table(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


NPB = 33      # tensors per Block in BlockGroupFn's flat argument list


def _ptrs5(tensors):
    return (C.c_void_p * 5)(*[t.data_ptr() for t in tensors])


class BlockGroupFn(torch.autograd.Function):
    """Up to three INDEPENDENT Blocks (dialog_models.py:253-271) as ONE forward launch and three backward launches
    (pv_block_fwd / pv_block_bwd): 16 batch rows per workgroup, activations in LDS, exact fp32 MFMA.  A Block alone
    occupies 16 of the 256 CUs, so the two pyramids' Blocks of one level and the three Blocks of a DLS_Auto level share
    launches.  Flat arguments, per Block: x [M, Din], gamma, beta, then inv_sigma, weight, bias, u, v, W^T of c1, c2, c3,
    c4, translate (5 each).  inv_sigma / u / v / W^T come from this step's spectral-norm pre-pass."""

    @staticmethod
    def forward(ctx, eps_list, *flat):
        n = len(flat) // NPB
        _need_cuda(*flat)
        dev = flat[0].device
        calls = (_lib.BlockCall * n)()
        saved, meta, outs = [], [], []
        for b in range(n):
            t = flat[b * NPB:(b + 1) * NPB]
            x, gamma, beta = t[0:3]
            inv, Ws, bs, us, vs, wts = t[3:8], t[8:13], t[13:18], t[18:23], t[23:28], t[28:33]
            din, mid, dout = Ws[0].shape[1], Ws[1].shape[0], Ws[4].shape[0]
            x2 = x.reshape(-1, din).contiguous()
            M = x2.shape[0]
            sw, gw, pw = C.c_int(0), C.c_int(0), C.c_int(0)
            _lib.call("pv_block_layout", din, mid, dout, C.byref(sw), C.byref(gw), C.byref(pw))
            Wc = [w.contiguous() for w in Ws]
            bc = [q.contiguous() for q in bs]
            g, be = gamma.contiguous(), beta.contiguous()
            save = torch.empty((M, sw.value), dtype=torch.float32, device=dev)
            mean = torch.empty(M, dtype=torch.float32, device=dev)
            rstd = torch.empty(M, dtype=torch.float32, device=dev)
            out = torch.empty((M, dout), dtype=torch.float32, device=dev)
            c = calls[b]
            c.M, c.Din, c.mid, c.Dout, c.ln_eps, c.accumulate = M, din, mid, dout, eps_list[b], 0
            c.W, c.Wt, c.bias, c.inv_sigma = _ptrs5(Wc), _ptrs5(wts), _ptrs5(bc), _ptrs5(inv)
            c.gamma, c.beta = g.data_ptr(), be.data_ptr()
            c.x, c.ldx, c.save, c.mean, c.rstd, c.out, c.ldo = (x2.data_ptr(), din, save.data_ptr(), mean.data_ptr(),
                                                                rstd.data_ptr(), out.data_ptr(), dout)
            saved += [x2, out, save, mean, rstd, g, *inv, *Wc, *bc, *us, *vs]
            meta.append((M, din, mid, dout, gw.value, pw.value, x.shape, (gamma, beta, tuple(Ws), tuple(bs))))
            outs.append(out.view(tuple(x.shape[:-1]) + (dout,)))
        _lib.call("pv_block_fwd", C.byref(calls), n, _stream())
        ctx.save_for_backward(*saved)
        ctx.meta = meta
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        meta = ctx.meta
        n = len(meta)
        st = ctx.saved_tensors
        per = len(st) // n
        dev = st[0].device
        live = [b for b in range(n) if douts[b] is not None]
        calls = (_lib.BlockCall * max(len(live), 1))()
        keep, results, all_fused = [], {}, True
        for k, b in enumerate(live):
            t = st[b * per:(b + 1) * per]
            x2, out, save, mean, rstd, g = t[:6]
            inv, Wc, bc, us, vs = t[6:11], t[11:16], t[16:21], t[21:26], t[26:31]
            M, din, mid, do, gw, pw, xshape, (p_g, p_b, p_W, p_bias) = meta[b]
            d2 = douts[b].reshape(M, do).contiguous()
            nwg = (M + 15) // 16
            dx = torch.empty((M, din), dtype=torch.float32, device=dev)
            dY = torch.empty((M, gw), dtype=torch.float32, device=dev)
            cpart = torch.empty(5 * 4 * nwg, dtype=torch.float32, device=dev)      # (a cluster member each: pevae_hip.h)
            ppart = torch.empty(((M + 3) // 4, pw), dtype=torch.float32, device=dev)      # a row per 4-row slab
            params = list(p_W) + list(p_bias) + [p_g, p_b]
            sinks = [grad_sink(p, tuple(p.shape)) for p in params]
            fused = all(s is not None for s in sinks)      # all-or-nothing: one accumulate flag per Block
            all_fused = all_fused and fused
            grads = sinks if fused else [torch.empty(tuple(p.shape), dtype=torch.float32, device=dev) for p in params]
            c = calls[k]
            c.M, c.Din, c.mid, c.Dout, c.ln_eps, c.accumulate = M, din, mid, do, 0.0, 1 if fused else 0
            c.W, c.bias, c.inv_sigma, c.u, c.v = _ptrs5(Wc), _ptrs5(bc), _ptrs5(inv), _ptrs5(us), _ptrs5(vs)
            c.gamma, c.beta = g.data_ptr(), None
            c.x, c.ldx, c.save, c.mean, c.rstd, c.out, c.ldo = (x2.data_ptr(), din, save.data_ptr(), mean.data_ptr(),
                                                                rstd.data_ptr(), out.data_ptr(), do)
            c.dout, c.lddo, c.dx, c.lddx = d2.data_ptr(), do, dx.data_ptr(), din
            c.dY, c.cpart, c.ppart = dY.data_ptr(), cpart.data_ptr(), ppart.data_ptr()
            c.dW, c.db = _ptrs5(grads[0:5]), _ptrs5(grads[5:10])
            c.dgamma, c.dbeta = grads[10].data_ptr(), grads[11].data_ptr()
            keep += [d2, dY, cpart, ppart]
            results[b] = (dx.view(xshape), None if fused else grads)
        if live:
            if ASYNC_WGRAD and all_fused and SIDE_BLOCK_WGRAD:
                # the dx chain now; the parameter gradients (full-chip GEMMs into the gradient sinks, nothing downstream reads
                # them) later on the side stream, beside the next Blocks' dx chains
                _lib.call("pv_block_bwd_phase", C.byref(calls), len(live), 1, _stream())
                flush_deferred()                    # the Blocks before this one: their GEMMs start beside this dx chain
                nlive, held = len(live), (calls, keep, list(st), [douts[b] for b in live])
                defer_side_work(lambda: _lib.call("pv_block_bwd_phase", C.byref(held[0]), nlive, 2, _stream()), keep=held)
            else:
                _lib.call("pv_block_bwd", C.byref(calls), len(live), _stream())
        ret = [None]
        for b in range(n):
            if b not in results:
                ret += [None] * NPB
                continue
            dx, grads = results[b]
            dW = [None] * 5 if grads is None else grads[0:5]
            db = [None] * 5 if grads is None else grads[5:10]
            dg, dbe = (None, None) if grads is None else (grads[10], grads[11])
            ret += [dx, dg, dbe] + [None] * 5 + list(dW) + list(db) + [None] * 15
        return tuple(ret)


def _block_args(blk):
    """BlockGroupFn's per-Block arguments after x (consumes the pre-pass results of the five linears)."""
    lins = (blk.c1, blk.c2, blk.c3, blk.c4, blk.translate)
    if any(m._inv_sigma_pre is None or getattr(m, "_wt_pre", None) is None for m in lins):
        # a Block called twice in one step (DeNAuto's discriminator: positive and negative pairs) runs its own power
        # iteration for the second call, as torch's spectral_norm hook does; the descriptor tables are cached on the Block,
        # so that nothing is uploaded once the step is being captured into a HIP graph
        # (a ring of four, indexed by the call's number within the step: the 1/sigma and W^T buffers of an earlier call
        # of this step are saved for ITS backward pass and must not be overwritten by the next call; the k-th extra call
        # of every step uses the same slot, so one eager step warms every slot a captured step will touch)
        ring = blk.__dict__.setdefault("_sn_ring", [{}, {}, {}, {}])
        blk.__dict__["_sn_turn"] = turn = (blk.__dict__.get("_sn_turn", -1) + 1) % len(ring)
        spectral_prepass(list(lins), blk.training, lins[0].eps, cache=ring[turn])
        us = [m.weight_u.clone() for m in lins]       # a later forward may advance the buffers before this backward runs
        vs = [m.weight_v.clone() for m in lins]
    elif getattr(lins[0], "clone_uv", False):
        blk.__dict__["_sn_turn"] = -1                 # first call of a step (it consumed the model's pre-pass)
        us = [m.weight_u.clone() for m in lins]
        vs = [m.weight_v.clone() for m in lins]
    else:
        blk.__dict__["_sn_turn"] = -1
        us = [m.weight_u for m in lins]
        vs = [m.weight_v for m in lins]
    inv, wts = [], []
    for m in lins:
        inv.append(m._inv_sigma_pre)
        wts.append(m._wt_pre)
        m._inv_sigma_pre = None
        m._wt_pre = None
    ln = blk.layer_norm
    return [ln.weight, ln.bias, *inv, *[m.weight_orig for m in lins], *[m.bias for m in lins], *us, *vs, *wts]


def block_forward(blk, x):
    """Run a dls_latent.Block through the fused kernels (the caller checked block_fusable)."""
    return BlockGroupFn.apply([blk.layer_norm.eps], x, *_block_args(blk))[0]


def blocks_forward(blks, xs):
    """Independent fusable Blocks side by side: one launch per pv_block_fwd group of up to three."""
    outs = []
    for i in range(0, len(blks), _lib.BLOCK_GROUP_MAX):
        grp, gx = blks[i:i + _lib.BLOCK_GROUP_MAX], xs[i:i + _lib.BLOCK_GROUP_MAX]
        flat = []
        for blk, x in zip(grp, gx):
            flat += [x] + _block_args(blk)
        outs += list(BlockGroupFn.apply([b.layer_norm.eps for b in grp], *flat))
    return outs


class DlsLevel(torch.autograd.Function):
    """One level of DLS_Auto's train branch after its three Blocks (dialog_models.py:3299-3310): posterior sample,
    residual adds and the two weighted detached-direction KLs in one launch each way.
    (Q, Pr [B,2n], a2, eps, lay [B,n]) -> (last_sample, output, ew * KL(t || q.detach) + cw * KL(t.detach || q))."""

    @staticmethod
    def forward(ctx, Q, Pr, a2, eps, lay, ew, cw):
        _need_cuda(Q, Pr, a2, eps, lay)
        Q, Pr, a2, eps, lay = [t.contiguous() for t in (Q, Pr, a2, eps, lay)]
        B, n = a2.shape
        dev = Q.device
        ls = torch.empty((B, n), dtype=torch.float32, device=dev)
        out = torch.empty((B, n), dtype=torch.float32, device=dev)
        kl = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.call("pv_dls_level_fwd", _p(Q), _p(Pr), _p(a2), _p(eps), _p(lay), B, n, ew, cw, _p(ls), _p(out), _p(kl),
                  _p(workspace("dls_level_kl_part", 256, torch.float32, Q.device)), _stream())
        ctx.save_for_backward(Q, Pr, eps)
        ctx.w = (ew, cw)
        ctx.set_materialize_grads(False)
        return ls, out, kl.view(())

    @staticmethod
    def backward(ctx, d_ls, d_out, g_kl):
        Q, Pr, eps = ctx.saved_tensors
        B, n = eps.shape
        ew, cw = ctx.w
        dQ, dPr = torch.empty_like(Q), torch.empty_like(Pr)
        da2, dlay = torch.empty_like(eps), torch.empty_like(eps)
        c = lambda t: None if t is None else t.contiguous()
        _lib.call("pv_dls_level_bwd", _p(Q), _p(Pr), _p(eps), _p(c(d_ls)), _p(c(d_out)), _p(c(g_kl)), B, n, ew, cw, _p(dQ),
                  _p(dPr), _p(da2), _p(dlay), _stream())
        return dQ, dPr, da2, None, dlay, None, None


class DlsLevelNext(torch.autograd.Function):
    """DlsLevel plus, in the same launch each way, the NEXT level's Block inputs (a1n, a2n given) and the running loss
    (rkl_in given): (Q, Pr, a2, eps, lay, a1n, a2n, rkl_in) -> (last_sample, output, rkl_in + kl term, q_in = [a1n + a2n |
    last_sample], p_in = [a2n | t_mean]).  Saves the DlsInputs launches, the scalar add per level and, backward, the chunk /
    zero-fill / accumulate launches autograd spends on t_mean's second consumer."""

    @staticmethod
    def forward(ctx, Q, Pr, a2, eps, lay, ew, cw, a1n, a2n, rkl_in):
        _need_cuda(Q, Pr, a2, eps, lay, a1n, a2n, rkl_in)
        Q, Pr, a2, eps, lay = [t.contiguous() for t in (Q, Pr, a2, eps, lay)]
        B, n = a2.shape
        dev = Q.device
        ls = torch.empty((B, n), dtype=torch.float32, device=dev)
        out = torch.empty((B, n), dtype=torch.float32, device=dev)
        kl = torch.empty(1, dtype=torch.float32, device=dev)
        wn, q_in, p_in = 0, None, None
        if a1n is not None:
            a1n, a2n = a1n.contiguous(), a2n.contiguous()
            wn = a1n.shape[1]
            q_in = torch.empty((B, wn + n), dtype=torch.float32, device=dev)
            p_in = torch.empty((B, wn + n), dtype=torch.float32, device=dev)
        add = None if rkl_in is None else rkl_in.detach().reshape(1).to(torch.float32)
        _lib.call("pv_dls_level_next_add_fwd", _p(Q), _p(Pr), _p(a2), _p(eps), _p(lay), B, n, ew, cw, _p(ls), _p(out), _p(kl),
                  _p(workspace("dls_level_kl_part", 256, torch.float32, dev)), _p(a1n), _p(a2n), wn, _p(q_in), _p(p_in), _p(add),
                  _stream())
        ctx.save_for_backward(Q, Pr, eps)
        ctx.w = (ew, cw, wn, rkl_in is not None)
        ctx.set_materialize_grads(False)
        return ls, out, kl.view(()), q_in, p_in

    @staticmethod
    def backward(ctx, d_ls, d_out, g_kl, dq_in, dp_in):
        Q, Pr, eps = ctx.saved_tensors
        B, n = eps.shape
        ew, cw, wn, has_add = ctx.w
        dQ, dPr = torch.empty_like(Q), torch.empty_like(Pr)
        da2, dlay = torch.empty_like(eps), torch.empty_like(eps)
        da1n = da2n = None
        if wn:
            da1n = torch.empty((B, wn), dtype=torch.float32, device=Q.device)
            da2n = torch.empty((B, wn), dtype=torch.float32, device=Q.device)
        c = lambda t: None if t is None else t.contiguous()
        _lib.call("pv_dls_level_next_bwd", _p(Q), _p(Pr), _p(eps), _p(c(d_ls)), _p(c(d_out)), _p(c(g_kl)), B, n, ew, cw, _p(dQ),
                  _p(dPr), _p(da2), _p(dlay), _p(c(dq_in)), _p(c(dp_in)), wn, _p(da1n), _p(da2n), _stream())
        return dQ, dPr, da2, None, dlay, None, None, da1n, da2n, (g_kl if has_add else None)


class DlsInputs(torch.autograd.Function):
    """q_in = [a1 + a2 | last_sample], p_in = [a2 | last_t] (dialog_models.py:3289-3297) in one launch."""

    @staticmethod
    def forward(ctx, a1, a2, ls, lt):
        _need_cuda(a1, a2, ls, lt)
        a1, a2, ls, lt = [t.contiguous() for t in (a1, a2, ls, lt)]
        B, w = a1.shape
        e = ls.shape[1]
        q_in = torch.empty((B, w + e), dtype=torch.float32, device=a1.device)
        p_in = torch.empty((B, w + e), dtype=torch.float32, device=a1.device)
        _lib.call("pv_dls_inputs_fwd", _p(a1), _p(a2), _p(ls), _p(lt), B, w, e, _p(q_in), _p(p_in), _stream())
        ctx.dims = (B, w, e)
        ctx.set_materialize_grads(False)
        return q_in, p_in

    @staticmethod
    def backward(ctx, dq, dp):
        B, w, e = ctx.dims
        dev = (dq if dq is not None else dp).device
        da1 = torch.empty((B, w), dtype=torch.float32, device=dev)
        da2 = torch.empty((B, w), dtype=torch.float32, device=dev)
        dls = torch.empty((B, e), dtype=torch.float32, device=dev)
        dlt = torch.empty((B, e), dtype=torch.float32, device=dev)
        c = lambda t: None if t is None else t.contiguous()
        _lib.call("pv_dls_inputs_bwd", _p(c(dq)), _p(c(dp)), B, w, e, _p(da1), _p(da2), _p(dls), _p(dlt), _stream())
        return da1, da2, dls, dlt


class L2NormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x2 = x.contiguous()
        rows, D = x2.shape
        a = torch.empty_like(x2)
        nrm = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.call("pv_l2norm_rows_fwd", _p(x2), rows, D, _p(a), _p(nrm), _stream())
        ctx.save_for_backward(a, nrm)
        return a

    @staticmethod
    def backward(ctx, da):
        a, nrm = ctx.saved_tensors
        dx = torch.empty_like(a)
        _lib.call("pv_l2norm_rows_bwd", _p(a), _p(nrm), _p(da.contiguous()), a.shape[0], a.shape[1], _p(dx), _stream())
        return dx


class InfoNCE(torch.autograd.Function):
    """(CE(log_softmax(S, rows), diag) + CE(log_softmax(S, cols), diag)) / 2 for a [B,B] score matrix."""

    @staticmethod
    def forward(ctx, S):
        _need_cuda(S)
        Sc = S.contiguous()
        B = Sc.shape[0]
        lr = torch.empty(B, dtype=torch.float32, device=S.device)
        lc = torch.empty(B, dtype=torch.float32, device=S.device)
        loss = torch.empty(1, dtype=torch.float32, device=S.device)
        _lib.call("pv_infonce_fwd", _p(Sc), B, _p(lr), _p(lc), _p(loss), _stream())
        ctx.save_for_backward(Sc, lr, lc)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        Sc, lr, lc = ctx.saved_tensors
        dS = torch.empty_like(Sc)
        _lib.call("pv_infonce_bwd", _p(Sc), _p(lr), _p(lc), _p(g.contiguous().view(1)), Sc.shape[0], _p(dS), _stream())
        return dS


class ScaledMatmulNT(torch.autograd.Function):
    """alpha * a b^T in fp32 (the InfoNCE score matrix v1 @ v2^T * 1.1, dialog_models.py:3271)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = a.contiguous(), b.contiguous()
        M, K = a.shape
        N = b.shape[0]
        c = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _lin(a, K, 1, b, 1, K, None, c, N, M, N, K, alpha=alpha)
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return c

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        M, K = a.shape
        N = b.shape[0]
        dc = dc.contiguous()
        da = torch.empty_like(a)
        db = torch.empty_like(b)
        _lin(dc, N, 1, b, K, 1, None, da, K, M, K, N, alpha=ctx.alpha)        # da = alpha dc b
        _lin(dc, 1, N, a, K, 1, None, db, K, N, K, M, alpha=ctx.alpha)        # db = alpha dc^T a
        return da, db, None


FAN_OUT = os.environ.get("PEVAE_FAN_OUT", "1") != "0"
FAN_MAX_TENSORS, FAN_MAX_USES = 8, 4          # PV_SUM_GROUP, PV_SUM_SRC of include/pevae_hip.h


class FanOut(torch.autograd.Function):
    """k aliases of each of up to 8 tensors, one per consumer: FanOut.apply(k, x0, x1, ...) -> (x0 alias 0..k-1, x1 alias
    0..k-1, ...).  No forward launch; the backward sums the gradients that arrived for each tensor in ONE launch
    (pv_sum_group) instead of autograd's one add per extra consumer.  Aliases nobody used cost nothing."""

    @staticmethod
    def forward(ctx, k, *xs):
        assert 1 <= k <= FAN_MAX_USES and 1 <= len(xs) <= FAN_MAX_TENSORS
        ctx.k = k
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        k = ctx.k
        outs, desc, nd = [], _lib.SumGroup(), 0
        for i in range(len(gs) // k):
            g = [t for t in gs[i * k:(i + 1) * k] if t is not None]
            if len(g) < 2:
                outs.append(g[0] if g else None)
                continue
            if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0
                       and t.data_ptr() % 16 == 0 for t in g):
                acc = g[0]
                for t in g[1:]:
                    acc = acc + t
                outs.append(acc)
                continue
            dst = torch.empty_like(g[0])
            desc.dst[nd] = dst.data_ptr()
            for j in range(FAN_MAX_USES):
                desc.src[nd][j] = g[j].data_ptr() if j < len(g) else None
            desc.n[nd] = dst.numel()
            nd += 1
            outs.append(dst)
        if nd:
            desc.count = nd
            _lib.call("pv_sum_group", C.byref(desc), _stream())
        return (None,) + tuple(outs)


class Fan(object):
    """Hands out FanOut aliases of a set of tensors, one per use: fan = Fan(k, tensors); fan(t) -> the next alias of t
    (t itself when it is not in the set or its aliases ran out: autograd then adds as usual)."""

    def __init__(self, k, tensors):
        live = []
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda and t.requires_grad and all(t is not u for u in live):
                live.append(t)
        live = live[:FAN_MAX_TENSORS]
        self.left = {}
        if FAN_OUT and live and torch.is_grad_enabled():
            al = FanOut.apply(k, *live)
            for i, t in enumerate(live):
                self.left[id(t)] = (t, list(al[i * k:(i + 1) * k]))

    def __call__(self, t):
        ent = self.left.get(id(t))
        if ent is None or ent[0] is not t or not ent[1]:
            return t
        return ent[1].pop(0)


class ConsLossSum(torch.autograd.Function):
    """sum_p cons_loss(v1_p, v2_p) for up to 7 pairs in one launch per stage (pv_cons_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, *vs):
        _need_cuda(*vs)
        vs = [v.contiguous() for v in vs]
        n, B = len(vs) // 2, vs[0].shape[0]
        dev = vs[0].device
        D = (C.c_int * n)(*[int(vs[2 * p].shape[1]) for p in range(n)])
        for p in range(n):
            assert vs[2 * p].shape == vs[2 * p + 1].shape and vs[2 * p].shape[0] == B and vs[2 * p].shape[1] % 4 == 0
        ws = torch.empty(_lib.load().pv_cons_loss_ws_floats(D, n, B), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        p1 = (C.c_void_p * n)(*[vs[2 * p].data_ptr() for p in range(n)])
        p2 = (C.c_void_p * n)(*[vs[2 * p + 1].data_ptr() for p in range(n)])
        _lib.call("pv_cons_loss_fwd", p1, p2, D, n, B, 1.1, _p(ws), _p(loss), _stream())
        ctx.ws, ctx.meta = ws, (n, B, D, [tuple(v.shape) for v in vs])
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        n, B, D, shapes = ctx.meta
        dev = ctx.ws.device
        outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
        d1 = (C.c_void_p * n)(*[outs[2 * p].data_ptr() for p in range(n)])
        d2 = (C.c_void_p * n)(*[outs[2 * p + 1].data_ptr() for p in range(n)])
        _lib.call("pv_cons_loss_bwd", _p(g.contiguous().view(1)), D, n, B, _p(ctx.ws), d1, d2, _stream())
        return tuple(outs)


def cons_loss_sum(pairs):
    """Sum of DLS_Auto.cons_loss over (v1, v2) pairs; batches of up to 7 pairs share their launches."""
    total = None
    for i in range(0, len(pairs), 7):
        flat = [t for pr in pairs[i:i + 7] for t in pr]
        part = ConsLossSum.apply(*flat)
        total = part if total is None else total + part
    return total


def cons_loss(v1, v2):
    """DLS_Auto.cons_loss (dialog_models.py:3266-3272) with batch_arrange = arange(B)."""
    return InfoNCE.apply(ScaledMatmulNT.apply(L2NormRows.apply(v1), L2NormRows.apply(v2), 1.1))
