"""ctypes binding of libmocogan_hip.so (declared in include/mocogan_hip.h).

PyTorch-ROCm supplies device memory and the HIP stream only; every call below hands raw
device pointers to a hand-written gfx950 kernel.  There is no fallback: if the library is
missing or a status is non-zero this module raises.
"""
import ctypes as C
import json
import os

import torch

from .build import lib_path

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3

_STATUS = {0: "MCG_OK", -1: "MCG_ERR_BAD_ARG", -2: "MCG_ERR_UNSUPPORTED", -3: "MCG_ERR_LAUNCH",
           -4: "MCG_ERR_WORKSPACE"}


class McgError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    """mcg_conv_geom"""
    _fields_ = [("N", C.c_int32), ("Ti", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ci", C.c_int32),
                ("To", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Co", C.c_int32),
                ("kt", C.c_int32), ("x_perm_n", C.c_int32), ("precision", C.c_int32),
                ("tile", C.c_int32), ("ci_valid", C.c_int32),
                ("x_stride0", C.c_int64), ("x_stride1", C.c_int64)]


SUMS_NONE, SUMS_STATS, SUMS_BN_BWD, SUMS_COL = 0, 1, 2, 3
IO_OUT_BF16, IO_Y_BF16, IO_G_BF16, IO_OUT_SPLIT = 1, 2, 4, 8          # MCG_IO_*: which tensors of an element-wise call are bf16


class ConvEpilogue(C.Structure):
    """mcg_conv_epilogue: what the conv's epilogue fuses (include/mocogan_hip.h)."""
    _fields_ = [("sums", C.c_int32), ("groups", C.c_int32), ("part", C.c_void_p), ("bn_y", C.c_void_p),
                ("bn_stats", C.c_void_p * 2), ("bn_act", C.c_int32), ("act", C.c_int32), ("addend", C.c_void_p * 2),
                ("sigma", C.c_float), ("seed", C.c_uint64), ("stream_id", C.c_uint64 * 2),
                ("mask_out", C.c_void_p), ("out_bf16", C.c_int32), ("mask_in", C.c_void_p), ("n_slots", C.c_int32),
                ("slot_stride", C.c_int32), ("bn_y_bf16", C.c_int32)]


_P, _I, _I64, _U64, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double
_GP = C.POINTER(ConvGeom)
_EP = C.POINTER(ConvEpilogue)

# name -> (restype, argtypes); mirrors include/mocogan_hip.h one to one
SIGNATURES = {
    "mcg_version": (_I, []),
    "mcg_conv_fprop": (_I, [_GP, _P, _P, _P, _P, _P]),
    "mcg_conv_dgrad": (_I, [_GP, _P, _P, _P, _P, _I, _I, _P]),
    "mcg_conv_wgrad": (_I, [_GP, _P, _P, _P, _P]),
    "mcg_conv_fprop_ex": (_I, [_GP, _P, _P, _P, _P, _EP, _P]),
    "mcg_conv_dgrad_ex": (_I, [_GP, _P, _P, _P, _P, _EP, _P]),
    "mcg_conv_epilogue_part_bytes": (_I64, [_GP, _I, _I]),
    "mcg_bn_stats_from_partials": (_I, [_I64, _I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P]),
    "mcg_bn_act_bwd_from_partials": (_I, [_I64, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "mcg_colsum_from_partials": (_I, [_I, _P, _I, _I, _P, _P, _P]),
    "mcg_randn_rowquad": (_I, [_I64, _I, _F, _U64, _U64, _P, _P]),
    "mcg_fc_fprop": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "mcg_fc_dgrad": (_I, [_I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "mcg_fc_wgrad": (_I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "mcg_bn_workspace_bytes": (_I64, [_I64, _I]),
    "mcg_bn_stats": (_I, [_I64, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P]),
    "mcg_bn_act_fwd": (_I, [_I64, _I, _I, _P, _I64, _I64, _P, _I, _P, _F, _U64, _U64, _P, _I, _P]),
    "mcg_bn_act_bwd": (_I, [_I64, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "mcg_bn_sums": (_I, [_I64, _I, _P, _P, _P, _P]),
    "mcg_bn_stats_from_sums": (_I, [_I64, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P]),
    "mcg_bn_bwd_sums": (_I, [_I64, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "mcg_bn_act_bwd_from_sums": (_I, [_I64, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mcg_colsum_acc": (_I, [_I64, _I, _P, _P, _P, _P]),
    "mcg_pack_clip": (_I, [_I, _I, _I, _I, _I, _P, _I64, _I64, _P, _F, _U64, _U64, _P, _P]),
    "mcg_unpack_clip": (_I, [_I, _I, _I, _I, _I, _P, _P, _P]),
    "mcg_pack_clip_u8": (_I, [_I, _I, _I, _I, _I, _P, _I64, _I64, _P, _F, _U64, _U64, _P, _P]),
    "mcg_concat_label_planes": (_I, [_I, _I64, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mcg_tanh_bwd_to_frames": (_I, [_I, _I, _I64, _P, _P, _P, _P]),
    "mcg_gru_seq_fwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mcg_gru_seq_bwd": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mcg_loss_dis": (_I, [_I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mcg_loss_gen": (_I, [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mcg_adam_wd": (_I, [_I64, _P, _P, _P, _P, _D, _D, _D, _D, _D, _D, _P, _P]),
    "mcg_randn": (_I, [_I64, _F, _U64, _U64, _P, _P]),
    "mcg_randint": (_I, [_I64, _I, _U64, _U64, _P, _P]),
    "mcg_split_planes": (_I, [_I64, _I64, _P, _P, _P]),
    "mcg_split_planes_multi": (_I, [_I, _P, _P]),
}

ABI_VERSION = 7          # MCG_ABI_VERSION of include/mocogan_hip.h these prototypes were written against

_lib = None


def load():
    """dlopen the in-tree library (never a site-packages copy) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise McgError("libmocogan_hip.so not built (%s): run __graft_entry__.build(); "
                       "there is no CPU fallback for the product path" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mcg_version() != ABI_VERSION:
        raise McgError("%s is ABI revision %d, this binding expects %d: rebuild it (__graft_entry__.build())"
                       % (path, lib.mcg_version(), ABI_VERSION))
    _lib = lib
    return lib


_tile_override = 0


def set_tile_override(t):
    """Tests / tuning tools: a default for mcg_conv_geom.tile of every conv call that does not set one itself
    (0 = none).  Lives here, in the caller: the library has no process-global state."""
    global _tile_override
    _tile_override = int(t)


def _with_override(g):
    if not _tile_override or g.tile:
        return g
    gg = ConvGeom.from_buffer_copy(g)
    gg.tile = _tile_override
    return gg


_SYNC_EVERY_CALL = os.environ.get("MCG_SYNC", "0") == "1"     # debugging aid: serialise host and device


def _check(status, name):
    if status != 0:
        raise McgError("%s failed: %s" % (name, _STATUS.get(status, status)))
    if _SYNC_EVERY_CALL:
        torch.cuda.synchronize()


def _p(t, dtype=torch.float32):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise McgError("the HIP path needs device tensors (got a %s tensor)" % t.device)
    if t.dtype != dtype:
        raise McgError("expected %s, got %s" % (dtype, t.dtype))
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dense(t):
    if t is not None and not t.is_contiguous():
        raise McgError("tensor must be dense")
    return t


PREC_F32, PREC_BF16, PREC_BF16_STORE, PREC_SPLIT, PREC_BF16_Y16 = 0, 1, 2, 3, 4
# 'bf16': bf16 MFMA on fp32 tensors (rounded in the kernel); 'bf16s': bf16 MFMA on operands that are bf16 in memory;
# 'f32x3': fp32 values as three bf16 terms (split_planes), six bf16 products per fp32 product -- fp32 results on the bf16 pipe
# 'bf16y': as 'bf16' with the y-side tensor bf16 in memory (x, w fp32): the clip-side layers of bf16 networks (wgrad, dgrad)
PRECISIONS = {"f32": PREC_F32, "fp32": PREC_F32, "bf16": PREC_BF16, "bf16s": PREC_BF16_STORE, "f32x3": PREC_SPLIT, "bf16y": PREC_BF16_Y16,
              PREC_F32: PREC_F32, PREC_BF16: PREC_BF16, PREC_BF16_STORE: PREC_BF16_STORE, PREC_SPLIT: PREC_SPLIT,
              PREC_BF16_Y16: PREC_BF16_Y16}


def _pin(g, t, side='x'):
    """device pointer of an INPUT operand of a conv launch (side: 'x' | 'y' | 'w'): bf16 tensors for MCG_PREC_BF16_STORE / split
    geometries, and for the y side of MCG_PREC_BF16_Y16"""
    if g.precision == PREC_BF16_Y16:
        return _p(t, torch.bfloat16 if side == 'y' else torch.float32)
    return _p(t, torch.bfloat16 if g.precision in (PREC_BF16_STORE, PREC_SPLIT) else torch.float32)


def _pany(t):
    """device pointer of a tensor that may be fp32 or bf16 (outputs of the element-wise passes) -> (pointer, is_bf16)"""
    if t.dtype == torch.bfloat16:
        return _p(t, torch.bfloat16), 1
    return _p(t), 0


def make_geom(N, Ti, Hi, Wi, Ci, Co, kt, x_stride0=None, x_perm_n=0, x_stride1=0, precision=PREC_F32, ci_valid=0):
    """Geometry of one k4 s(1,2,2) p(0,1,1) layer; x side [N][Ti][Hi][Wi][Ci], y side dense.  ci_valid: channels of x
    that carry data (0 = all Ci): 3 for the RGB clip stored with Ci = 4."""
    g = ConvGeom()
    g.precision = PRECISIONS[precision]
    g.ci_valid = ci_valid
    g.N, g.Ti, g.Hi, g.Wi, g.Ci = N, Ti, Hi, Wi, Ci
    g.To, g.Ho, g.Wo, g.Co, g.kt = Ti - kt + 1, Hi // 2, Wi // 2, Co, kt
    g.x_perm_n = x_perm_n
    g.x_stride0 = Ti * Hi * Wi * Ci if x_stride0 is None else x_stride0
    g.x_stride1 = x_stride1
    return g


# ------------------------------------------------------------------------------------------
# optional per-launch timing of the conv kernels with HIP events on the launch stream (bench.py's
# roofline leg).  Off by default; costs two event records per conv launch when on.
# ------------------------------------------------------------------------------------------
_timing = None
_tag = ""


def set_tag(tag):
    """Label (network name) attached to the conv launches that follow."""
    global _tag
    _tag = tag


def get_tag():
    return _tag


def timing_begin():
    global _timing
    _timing = {}


def timing_end():
    """-> {"<tag>.<pass>": (launches, total_ms)}; synchronises the device."""
    global _timing
    t, _timing = _timing, None
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in (t or {}).items()}


split_launches = 0          # conv launches in the MCG_PREC_SPLIT form so far (tests assert that the form really ran)
split_only_outputs = 0      # element-wise launches that wrote their output in the split layout only


def _launch(kind, fn, *args):
    if args[0].precision == PREC_SPLIT:
        global split_launches
        split_launches += 1
    if _timing is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*args)
    e1.record()
    _timing.setdefault(_tag + "." + kind, []).append((e0, e1))
    g = args[0]                                             # per-geometry detail: "<tag>.<pass> N T H Ci Co"
    _timing.setdefault("%s.%s N=%d T=%d H=%d Ci=%d Co=%d" % (_tag, kind, g.N, g.Ti, g.Hi, g.Ci, g.Co), []).append((e0, e1))
    return r


# ------------------------------------------------------------------------------------------
# per-geometry choice of the GEMM block tile (mcg_conv_geom.tile).  The best of the six candidates
# depends on how the tile count of a launch divides over the 256 CUs and, for dgrad, on how many blocks
# skip dead temporal taps -- the library's closed-form heuristic misses by up to 20 % on some layers.
# With autotune on, the first launch of each (pass, geometry) times every candidate once on scratch
# tensors and the winner is used from then on.  Off by default (tests, parity runs): bench.py, train.py
# and generate_samples.py switch it on.
# ------------------------------------------------------------------------------------------
TILE_CANDIDATES = (0, 101, 102, 103, 201, 202, 203)     # (the long tiles 4 = 256x64 and 5 = 64x256 exist, but when they
                                                        # win the isolated timing they lose inside the iteration: measured)
FPROP_SPLIT_CANDIDATES = (1103, 1203, 1202, 2103, 2203, 2202)     # 2- / 4-way split-K: only when few tiles (see _tuned)
WGRAD_SPLIT_CANDIDATES = (2007, 2008, 2010, 1010)      # LDS-DMA weight gradient with half / twice the pixel splits (few-tile layers)
V2_CANDIDATES = (7, 8, 10)                              # gemm_bf16_v2_kernel (bf16-stored / split / fp32 operands): 256x128 / 256x256 (wgrad 128x256 / 256x256) one block per CU; 10 = 128x128, two blocks per CU
_autotune = False
_tile_cache = {}


_PRETUNED = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_tiles_mi355x.json')


def set_autotune(on, use_pretuned=True):
    """on: time the tile candidates at the first launch of every (pass, geometry) not known yet.  use_pretuned: start
    from the table shipped with the package -- the choices this tuner made on an MI355X for the reference's layer
    geometries at batch 32 per GPU (tuned_tiles_mi355x.json); anything else is still tuned on first use."""
    global _autotune
    _autotune = bool(on)
    if use_pretuned and os.path.exists(_PRETUNED) and os.environ.get('MCG_NO_PRETUNED') != '1':
        for k, v in json.load(open(_PRETUNED)):
            # tuner off (train.py --autotune 0): the tile heuristic replaces the table's tile codes, but WHICH FORM a launch of an
            # 'f32x3' network takes (split on the bf16 pipe / fp32 MFMA) has no heuristic -- those entries are loaded either way,
            # otherwise such a network would silently run every GEMM on the fp32 kernels (round 4's advice)
            if on or str(k[0]).startswith('split-'):
                _tile_cache.setdefault(tuple(k), int(v))


def reset_tuning():
    """Back to the state of a fresh import: tuner off, no cached tile choices, no tile override.  (tests/conftest.py calls this
    before every test module so that no module inherits a timing-dependent tile table from an earlier one; train.main() and
    bench.py switch the tuner on for their own process.)"""
    global _autotune, _tile_override
    _autotune = False
    _tile_override = 0
    _tile_cache.clear()


def tile_choices():
    """{(pass, geometry...): tile code} chosen so far (for logs / DESIGN.md tables)."""
    return dict(_tile_cache)


def save_tile_choices(path):
    """Persist the tuned choices (JSON) so that a later process -- or a profiler pass that must not see the
    tuning launches -- can start from them."""
    import json
    with open(path, 'w') as f:
        json.dump([[list(k), v] for k, v in _tile_cache.items()], f)


def load_tile_choices(path):
    import json
    for k, v in json.load(open(path)):
        _tile_cache[tuple(k)] = int(v)


def with_precision(g, precision):
    """a copy of the geometry for another operand form of the same layer (its tile is chosen afresh)"""
    h = ConvGeom.from_buffer_copy(g)
    h.precision = PRECISIONS[precision]
    h.tile = 0
    return h


def split_covers(kind, g):
    """does the MCG_PREC_SPLIT form of this pass exist for the geometry (the library's own condition, restated for callers)?
    kind 'fprop': the sum runs over Ci; 'dgrad': over Co, and the LDS-DMA input-gradient tiles need >= 64 output columns;
    'wgrad': the sum runs over pixels, x and y both in the split layout."""
    def p2(c, lo):
        return c >= lo and c & (c - 1) == 0
    x_el = max((g.N - 1) * g.x_stride0, 0) + g.Ti * g.Hi * g.Wi * g.Ci if not g.x_perm_n else None
    y_el = g.N * g.To * g.Ho * g.Wo * g.Co
    w_el = g.Co * g.kt * 16 * g.Ci
    if w_el * 8 >= 1 << 31:
        return False
    if kind == 'fprop':
        return p2(g.Ci, 16) and x_el is not None and x_el * 8 < 1 << 31
    if kind == 'wgrad':                                             # both operands split; the LDS-DMA weight-gradient tiles' own limits
        return (g.Co >= 128 and g.Co % 64 == 0 and p2(g.Ci, 64) and x_el is not None and x_el * 8 < 1 << 31 and y_el * 8 < 1 << 31
                and (y_el // g.Co) % 16 == 0)
    return p2(g.Co, 16) and p2(g.Ci, 64) and y_el * 8 < 1 << 31


def split_decided(kind, g):
    """True when a launch of this pass and geometry is KNOWN to take the split form (MCG_SPLIT=always, or the table says so): the
    producers of its operands may then write the split layout only.  False while undecided."""
    if not split_covers(kind, g):
        return False
    mode = os.environ.get('MCG_SPLIT', 'auto')
    if mode != 'auto':
        return mode == 'always'
    return _tile_cache.get(_geom_key('split-' + kind, g)) == 1


_warned_undecided_split = False


def split_pays(kind, g, run_plain, run_split):
    """Which of the two forms of a launch is faster for this geometry -- the fp32-MFMA kernels on fp32 operands (run_plain) or the
    split form on the bf16 pipe (run_split, INCLUDING whatever it takes to produce the split operands)?  Timed once per (pass,
    geometry) like the tile candidates and kept in the same table (key 'split-<pass>', value 1 = split); MCG_SPLIT=always / never
    overrides (tests, A/B timing).  Both forms write the same output; the caller launches the winner afterwards."""
    mode = os.environ.get('MCG_SPLIT', 'auto')
    if mode != 'auto':
        return mode == 'always'
    key = _geom_key('split-' + kind, g)
    c = _tile_cache.get(key)
    if c is None and not _autotune:
        # tuner off (tests, train.py --autotune 0): nothing is timed -- the table's entries, else the fp32 form.
        # (Under data parallelism per-rank timing could also leave the ranks on different forms.)
        global _warned_undecided_split
        if not _warned_undecided_split:
            _warned_undecided_split = True
            import warnings
            warnings.warn("precision 'f32x3': the tile tuner is off and the table holds no decision for %s N=%d T=%d H=%d Ci=%d Co=%d -- "
                          "this launch (and every other undecided one) runs the fp32-MFMA form; hiplib.set_autotune(True), "
                          "MCG_SPLIT=always or a loaded tile table decide it" % (kind, g.N, g.Ti, g.Hi, g.Ci, g.Co))
        return False
    if c is None:
        global _timing
        saved, _timing = _timing, None
        try:
            best = {}
            for name, fn in (('plain', run_plain), ('split', run_split)):
                fn()
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    fn()
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1)
                    best[name] = ms if name not in best else min(best[name], ms)
            c = int(best['split'] < 0.97 * best['plain'])
        finally:
            _timing = saved
        _tile_cache[key] = c
    return bool(c)


def _geom_key(kind, g, extra=()):
    return (kind, g.N, g.Ti, g.Hi, g.Wi, g.Ci, g.Co, g.kt, g.x_perm_n, g.precision) + tuple(extra)


def _tuned(kind, g, extra, out_side, run_on):
    """Returns a copy of g with .tile set to the fastest candidate.  run_on(geom, scratch) launches the pass
    with its OUTPUT directed to `scratch` (extent of the geometry's `out_side`), so tuning never touches
    the caller's output or accumulators."""
    if not _autotune or g.tile:
        return g
    key = _geom_key(kind, g, extra)
    code = _tile_cache.get(key)
    if code is None:
        global _timing
        saved, _timing = _timing, None                      # tuning launches are not part of any measurement
        cur = torch.cuda.current_stream()
        best, code = None, 0
        gg = ConvGeom.from_buffer_copy(g)
        scratch = _scratch_like(g, out_side)

        def run(geom):
            run_on(geom, scratch)
        try:
            cands = TILE_CANDIDATES
            if (kind == "dgrad" and 4 < g.Ci <= 64) or (kind == "fprop" and g.Co <= 64 and g.Ci > 4):
                cands = cands + (4,)                                # 256 x 64: the widest tile a 64-column output admits
            if kind == "wgrad" and g.Ci == 4:
                cands = cands + (6,)                                # patch-in-LDS kernel (also what tile 0 selects when it applies)
            out_elems = g.N * g.To * g.Ho * g.Wo * g.Co if kind == "fprop" else g.N * g.Ti * g.Hi * g.Wi * g.Ci
            if kind in ("fprop", "dgrad") and g.Ci > 4 and out_elems <= (1 << (23 if kind == "fprop" else 25)):
                cands = cands + FPROP_SPLIT_CANDIDATES          # <= 1024 tiles of 64x64: K splits can fill the CUs
            if g.precision == PREC_BF16_STORE and g.Ci >= 64 and g.Co >= 64:
                # the LDS-DMA kernels (the library refuses what they do not cover).  fp32 networks can run them too (tile codes
                # 7 / 8 by request), but measured on the MI355X they do not beat the register-staged fp32 kernels: 110-127
                # against 128-134 TFLOP/s on the big layers, and one 256-row block per CU quantises badly at batch 32
                if not (kind == "dgrad" and g.Ci == 64):        # (64 output columns per parity class: the 256x64 tile never wins)
                    cands = cands + V2_CANDIDATES
                    if kind == "wgrad" and g.Co * g.kt * 16 * g.Ci <= (1 << 21):
                        cands = cands + WGRAD_SPLIT_CANDIDATES
                else:
                    cands = cands + (10,)                       # ... but 256x64 with two buffers lets two blocks share a CU
                    if g.Ho == 16 and g.Wo == 16:
                        cands = cands + (9,)                    # ... and the patch-stationary kernel, four classes per block, is made for it
            if g.precision == PREC_F32 and g.Ci >= 64 and g.Co >= 64 and kind != "dgrad" or (g.precision == PREC_F32 and kind == "dgrad" and g.Ci >= 128 and g.Co >= 64):
                cands = cands + (10,)                           # the LDS-DMA kernel on fp32 operands, 128x128, two blocks per CU
            if g.precision == PREC_SPLIT:
                # (the LDS-DMA kernels are the only ones that multiply split operands; one 256-row block per CU: late layers need K splits)
                cands = V2_CANDIDATES + ((1007, 2007, 1010, 2010) if kind in ("fprop", "dgrad") and out_elems <= (1 << 25) else ())
                if kind == "wgrad" and g.Co * g.kt * 16 * g.Ci <= (1 << 21):
                    # few (Co, tap x Ci) tiles -- the 2-D layers: many pixel splits then add onto the same small dw with float
                    # atomics; + 2000 halves / + 1000 doubles the number of splits (round 6)
                    cands = cands + WGRAD_SPLIT_CANDIDATES
                if kind == "dgrad" and g.Ci == 64 and g.Ho == 16 and g.Wo == 16:
                    cands = cands + (9,)                        # the patch-stationary kernel (four parity classes per block)
            for cand in cands:
                gg.tile = cand
                try:
                    run(gg)                                 # warm-up (and rejects impossible candidates)
                except McgError:
                    continue
                ms = None
                for _ in range(3):                          # best of three timings of two launches each
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    run(gg)
                    run(gg)
                    e1.record(cur)
                    e1.synchronize()
                    t = e0.elapsed_time(e1)
                    ms = t if ms is None or t < ms else ms
                if best is None or ms < best:
                    best, code = ms, cand
        finally:
            _timing = saved
        _tile_cache[key] = code
    if not code:
        return g
    gg = ConvGeom.from_buffer_copy(g)
    gg.tile = code
    return gg


def _scratch_like(g, which):
    """dense scratch tensors with the extents of the geometry's x / y / w sides"""
    if which == 'x':
        if g.x_perm_n:
            n = (g.x_perm_n - 1) * g.x_stride0 + (g.N // g.x_perm_n - 1) * g.x_stride1 + g.Ti * g.Hi * g.Wi * g.Ci
        else:
            n = (g.N - 1) * g.x_stride0 + g.Ti * g.Hi * g.Wi * g.Ci
        return torch.zeros(n, device='cuda')
    if which == 'y':
        return torch.zeros(g.N * g.To * g.Ho * g.Wo * g.Co, device='cuda')
    return torch.zeros(g.Co * g.kt * 16 * g.Ci, device='cuda')


# ------------------------------------------------------------------------------------------
# thin typed wrappers (tensors in; nothing allocated here except the one-off tuning scratch)
# ------------------------------------------------------------------------------------------
def _fprop(g, x, w, bias, y):
    if y.dtype == torch.bfloat16:                     # bf16 output: the flag travels in an (otherwise empty) epilogue
        return _fprop_ex(g, x, w, bias, y, epilogue(out_bf16=True), split_ok=True)
    g = _with_override(g)
    _check(load().mcg_conv_fprop(C.byref(g), _pin(g, x), _pin(g, _dense(w), 'w'), _p(bias), _p(_dense(y)), _stream()), "mcg_conv_fprop")


def _dgrad(g, y, w, bias, x, act, accumulate):
    if x.dtype == torch.bfloat16:
        assert act == ACT_NONE and not accumulate
        return _dgrad_ex(g, y, w, bias, x, epilogue(out_bf16=True), split_ok=True)
    g = _with_override(g)
    _check(load().mcg_conv_dgrad(C.byref(g), _pin(g, _dense(y), 'y'), _pin(g, _dense(w), 'w'), _p(bias), _p(x), act, int(accumulate), _stream()),
           "mcg_conv_dgrad")


def _wgrad(g, x, y, dw):
    g = _with_override(g)
    _check(load().mcg_conv_wgrad(C.byref(g), _pin(g, x), _pin(g, _dense(y), 'y'), _p(_dense(dw)), _stream()), "mcg_conv_wgrad")


# ---- fused epilogues (mcg_conv_epilogue) ---------------------------------------------------------
def _vp(t, dtype=torch.float32):
    """raw device address (0 for None) for the pointer fields of ConvEpilogue"""
    p = _p(t, dtype)
    return None if p is None else p.value


def epilogue(sums=SUMS_NONE, groups=1, part=None, bn_y=None, bn_stats=(None, None), bn_act=ACT_NONE, act=ACT_NONE,
             addend=(None, None), sigma=0.0, seed=0, stream_id=(0, 0), mask_out=None, mask_in=None, out_bf16=False, out_split=False):
    """Builds a ConvEpilogue; the tensors must stay alive until the launch has been queued (they are the caller's)."""
    ep = ConvEpilogue()
    y16 = bn_y is not None and bn_y.dtype == torch.bfloat16
    ep.sums, ep.groups, ep.part, ep.bn_y = sums, groups, _vp(part), _vp(_dense(bn_y), torch.bfloat16 if y16 else torch.float32)
    ep.bn_y_bf16 = int(y16)
    ep.bn_stats[0], ep.bn_stats[1] = _vp(bn_stats[0]), _vp(bn_stats[1] if len(bn_stats) > 1 else None)
    ep.bn_act, ep.act = bn_act, act
    ep.addend[0], ep.addend[1] = _vp(_dense(addend[0])), _vp(_dense(addend[1] if len(addend) > 1 else None))
    ep.sigma, ep.seed = float(sigma), int(seed)
    ep.stream_id[0], ep.stream_id[1] = int(stream_id[0]), int(stream_id[1] if len(stream_id) > 1 else 0)
    ep.mask_out, ep.mask_in = _vp(_dense(mask_out), torch.int32), _vp(_dense(mask_in), torch.int32)
    ep.out_bf16 = IO_OUT_SPLIT if out_split else int(bool(out_bf16))      # (out_split: the MCG_PREC_SPLIT layout, a bf16 tensor of 4 * Co columns)
    return ep


def epilogue_part_floats(g, kind, groups):
    """floats the per-tile partial sums of a fused epilogue can need for geometry g ('fprop' | 'dgrad')"""
    return int(load().mcg_conv_epilogue_part_bytes(C.byref(g), 0 if kind == "fprop" else 1, groups)) // 4


def _no_split(g):
    """the tile code without its split-K part: partial tiles cannot carry an epilogue"""
    if g.tile < 1000:
        return g
    gg = ConvGeom.from_buffer_copy(g)
    gg.tile = g.tile % 1000
    return gg


def _fprop_ex(g, x, w, bias, y, ep, split_ok=False):
    """split_ok: leave a split-K tile code in place (the library then refuses it for a bf16 output: the caller asked
    fprop_tile / dgrad_tile first)"""
    g = _with_override(g) if split_ok else _no_split(_with_override(g))
    assert bool(ep.out_bf16) == (y.dtype == torch.bfloat16)
    yp = _p(_dense(y), torch.bfloat16) if ep.out_bf16 else _p(_dense(y))
    _check(load().mcg_conv_fprop_ex(C.byref(g), _pin(g, x), _pin(g, _dense(w), 'w'), _p(bias), yp, C.byref(ep), _stream()), "mcg_conv_fprop_ex")


def _dgrad_ex(g, y, w, bias, x, ep, split_ok=False):
    g = _with_override(g) if split_ok else _no_split(_with_override(g))
    assert bool(ep.out_bf16) == (x.dtype == torch.bfloat16)
    xp = _p(_dense(x), torch.bfloat16) if ep.out_bf16 else _p(_dense(x))
    _check(load().mcg_conv_dgrad_ex(C.byref(g), _pin(g, _dense(y), 'y'), _pin(g, _dense(w), 'w'), _p(bias), xp, C.byref(ep), _stream()),
           "mcg_conv_dgrad_ex")


def dgrad_c4_mfma_covers(g):
    """True when conv_dgrad(g, ..., bias=None, act=ACT_NONE) runs the MFMA col2im kernel of the Ci = 4 layers (the library's
    own condition, restated for callers that split bias + tanh off into an element-wise pass to reach that kernel)."""
    frame = g.Ti * g.Hi * g.Wi * g.Ci
    whole = (g.x_stride1 == frame and g.x_stride0 == (g.N // g.x_perm_n) * frame) if g.x_perm_n else g.x_stride0 == frame
    return (g.Ci == 4 and 0 < g.ci_valid <= 3 and g.Co == 64 and g.Wo in (16, 32) and g.Ho % (128 // g.Wo) == 0 and whole
            and g.precision not in (PREC_BF16_STORE, PREC_SPLIT) and _with_override(g).tile in (0, 6))


def fprop_tile(g, x, w, bias):
    """the tile code the PLAIN conv_fprop uses for this geometry (tuned now if it has to be): codes >= 1000 split K, and a
    split launch can neither carry an epilogue nor write a bf16 output -- callers that want either ask before they allocate.
    (bf16 networks: the fused / bf16-output form is tuned separately and only over the tiles that admit it.)"""
    if _autotune and not g.tile:
        probe = torch.empty(0, device='cuda')
        g = _tuned("fprop", g, _ep_key(g, None, probe), 'y', _tune_run("fprop", g, x, w, bias, probe, None))
    return _with_override(g).tile


def dgrad_tile(g, y, w, bias, act=ACT_NONE, accumulate=False):
    if _autotune and not g.tile:
        probe = torch.empty(0, device='cuda')
        g = _tuned("dgrad", g, (act, int(accumulate)) + _ep_key(g, None, probe), 'x', _tune_run("dgrad", g, y, w, bias, probe, None, act, accumulate))
    return _with_override(g).tile


def _ep_key(g, ep, out):
    """bf16 networks tune a geometry per launch FORM: a fused epilogue / a bf16 output change which tile wins (the LDS-DMA
    kernels' row-wise epilogue costs differently from the register-staged kernels'), so the form is part of the key and the
    candidates are timed in it.  fp32 geometries keep their single key (and the shipped table)."""
    if g.precision != PREC_BF16_STORE:
        return ()
    return ('ep', int(ep.sums) if ep is not None else 0, int(bool(ep.mask_in)) if ep is not None else 0, int(out.dtype == torch.bfloat16))


def _tune_run(kind, g, a, w, bias, out_like, ep, act=ACT_NONE, accumulate=False):
    """the launch a tuning candidate is timed with: the caller's form (epilogue, output type) on scratch output"""
    def run(gg, scratch):
        out = scratch.view(torch.bfloat16)[:scratch.numel()] if out_like.dtype == torch.bfloat16 else scratch
        if g.precision == PREC_BF16_STORE and (ep is not None or out_like.dtype == torch.bfloat16) and gg.tile < 1000:
            e2 = ep if ep is not None else epilogue(out_bf16=True)
            (_fprop_ex if kind == "fprop" else _dgrad_ex)(gg, a, w, bias, out, e2, split_ok=True)
        elif out_like.dtype == torch.bfloat16:
            raise McgError("a split-K tile cannot write a bf16 output")
        elif kind == "fprop":
            _fprop(gg, a, w, bias, scratch)
        else:
            _dgrad(gg, a, w, bias, scratch, act, accumulate)
    return run


def conv_fprop(g, x, w, bias, y, ep=None, must_fuse=False):
    """ep: a ConvEpilogue to fuse into the launch.  Returns True when it was fused; False when the tile tuned for this
    geometry splits K (partial tiles cannot carry an epilogue) and must_fuse is off: then the PLAIN convolution ran and
    the caller does the epilogue's work with the stand-alone passes.  must_fuse drops the split instead."""
    if _autotune and not g.tile:
        g = _tuned("fprop", g, _ep_key(g, ep, y), 'y', _tune_run("fprop", g, x, w, bias, y, ep))   # x, w are only read
    if ep is not None and (must_fuse or _with_override(g).tile < 1000):
        _launch("fprop", _fprop_ex, g, x, w, bias, y, ep)
        return True
    _launch("fprop", _fprop, g, x, w, bias, y)
    return False


def conv_dgrad(g, y, w, bias, x, act=ACT_NONE, accumulate=False, ep=None, must_fuse=False):
    """ep / return value as in conv_fprop (ep needs act == ACT_NONE, no accumulate, dense x)."""
    if _autotune and not g.tile:
        g = _tuned("dgrad", g, (act, int(accumulate)) + _ep_key(g, ep, x), 'x', _tune_run("dgrad", g, y, w, bias, x, ep, act, accumulate))
    if ep is not None and (must_fuse or _with_override(g).tile < 1000):
        assert act == ACT_NONE and not accumulate
        _launch("dgrad", _dgrad_ex, g, y, w, bias, x, ep)
        return True
    _launch("dgrad", _dgrad, g, y, w, bias, x, act, accumulate)
    return False


def conv_wgrad(g, x, y, dw):
    if _autotune and not g.tile:
        g = _tuned("wgrad", g, (), 'w', lambda gg, out: _wgrad(gg, x, y, out))
    _launch("wgrad", _wgrad, g, x, y, dw)


def fc_fprop(M, K, Co, x, w, bias, y):
    _check(load().mcg_fc_fprop(M, K, Co, _p(_dense(x)), _p(_dense(w)), _p(bias), _p(_dense(y)), _stream()), "mcg_fc_fprop")


def fc_dgrad(M, K, Co, y, w, bias, bias_period, x):
    _check(load().mcg_fc_dgrad(M, K, Co, _p(_dense(y)), _p(_dense(w)), _p(bias), bias_period, _p(_dense(x)), _stream()), "mcg_fc_dgrad")


def fc_wgrad(M, K, Co, x, y, dw, db=None):
    _check(load().mcg_fc_wgrad(M, K, Co, _p(_dense(x)), _p(_dense(y)), _p(_dense(dw)), _p(db), _stream()), "mcg_fc_wgrad")


def bn_workspace_floats(C_max):
    return int(load().mcg_bn_workspace_bytes(0, C_max)) // 4


def bn_stats(M, Cn, y, gamma, beta, stats, avg_mean, avg_var, ws, eps=2e-5, decay=0.9, sync=None):
    """sync: None, or an object with .world and .all_reduce_sum(tensor) (step.GradExchange): the statistics are then
    those of the global batch (every rank holds M rows)."""
    if sync is None or sync.world == 1:
        _check(load().mcg_bn_stats(M, Cn, _p(_dense(y)), _p(gamma), _p(beta), _p(stats), _p(avg_mean), _p(avg_var),
                                   eps, decay, _p(ws), _stream()), "mcg_bn_stats")
        return
    sums = torch.empty(2 * Cn, dtype=torch.float64, device=y.device)
    _check(load().mcg_bn_sums(M, Cn, _p(_dense(y)), _p(sums, torch.float64), _p(ws), _stream()), "mcg_bn_sums")
    sync.all_reduce_sum(sums)
    _check(load().mcg_bn_stats_from_sums(M * sync.world, Cn, _p(sums, torch.float64), _p(gamma), _p(beta), _p(stats), _p(avg_mean),
                                         _p(avg_var), eps, decay, _stream()), "mcg_bn_stats_from_sums")


def _pout(out, split_out):
    """(pointer, MCG_IO_OUT_* flag) of an element-wise output: fp32, bf16, or (split_out) the split layout of PREC_SPLIT"""
    if split_out:
        global split_only_outputs
        split_only_outputs += 1
        return _p(_dense(out), torch.bfloat16), IO_OUT_SPLIT
    op, o16 = _pany(_dense(out))
    return op, IO_OUT_BF16 * o16


def bn_act_fwd(M, Cn, y, scale_shift, act, out, addend=None, sigma=0.0, seed=0, stream_id=0, c_valid=None,
               rows_per_item=0, item_stride=0, split_out=False):
    """y dense [M][Cn], or (rows_per_item > 0) a view whose items are item_stride elements apart.
    split_out: `out` is [M][4 Cn] bf16 and receives the split layout (split_planes' form) instead of fp32 values."""
    if rows_per_item == 0:
        _dense(y)
    op, oflag = _pout(out, split_out)
    yp, y16 = _pany(y)
    _check(load().mcg_bn_act_fwd(M, Cn, Cn if c_valid is None else c_valid, yp, rows_per_item, item_stride, _p(scale_shift), act,
                                 _p(_dense(addend)), sigma, seed, stream_id, op, oflag + IO_Y_BF16 * y16, _stream()),
           "mcg_bn_act_fwd")


def bn_act_bwd(M, Cn, g_out, y, stats, gamma, act, gx, dgamma, dbeta, ws, sync=None, split_out=False):
    gp, gflag = _pout(gx, split_out)
    g16 = gflag == IO_OUT_BF16
    ip, i16 = _pany(_dense(g_out))
    yp, y16 = _pany(_dense(y))
    if sync is None or sync.world == 1 or stats is None:
        _check(load().mcg_bn_act_bwd(M, Cn, ip, yp, _p(stats), _p(gamma), act, gp, gflag + IO_Y_BF16 * y16 + IO_G_BF16 * i16,
                                     _p(dgamma), _p(dbeta), _p(ws), _stream()), "mcg_bn_act_bwd")
        return
    assert not split_out
    local = torch.empty(2 * Cn, dtype=torch.float64, device=y.device)
    _check(load().mcg_bn_bwd_sums(M, Cn, ip, yp, _p(stats), act, IO_Y_BF16 * y16 + IO_G_BF16 * i16, _p(local, torch.float64), _p(ws),
                                  _stream()), "mcg_bn_bwd_sums")
    glob = local.clone()
    sync.all_reduce_sum(glob)
    _check(load().mcg_bn_act_bwd_from_sums(M, M * sync.world, Cn, ip, yp, _p(stats), _p(gamma), act, _p(local, torch.float64),
                                           _p(glob, torch.float64), gp, IO_OUT_BF16 * g16 + IO_Y_BF16 * y16 + IO_G_BF16 * i16,
                                           _p(dgamma), _p(dbeta), _p(ws), _stream()), "mcg_bn_act_bwd_from_sums")


def bn_stats_from_partials(M, Cn, part, n_slots, slot_stride, gamma, beta, stats, avg_mean, avg_var, ws, eps=2e-5, decay=0.9):
    """part: the (group's) partial sums a fused conv epilogue wrote (SUMS_STATS)."""
    _check(load().mcg_bn_stats_from_partials(M, Cn, _p(part), n_slots, slot_stride, _p(gamma), _p(beta), _p(stats), _p(avg_mean),
                                             _p(avg_var), eps, decay, _p(ws), _stream()), "mcg_bn_stats_from_partials")


def bn_act_bwd_from_partials(M, Cn, g_out, y, stats, gamma, act, part, n_slots, slot_stride, gx, dgamma, dbeta, ws, split_out=False):
    gp, gflag = _pout(gx, split_out)
    ip, i16 = _pany(_dense(g_out))
    yp, y16 = _pany(_dense(y))
    _check(load().mcg_bn_act_bwd_from_partials(M, Cn, ip, yp, _p(stats), _p(gamma), act, _p(part), n_slots, slot_stride, gp,
                                               gflag + IO_Y_BF16 * y16 + IO_G_BF16 * i16, _p(dgamma), _p(dbeta), _p(ws),
                                               _stream()), "mcg_bn_act_bwd_from_partials")


def colsum_from_partials(Cn, part, n_slots, slot_stride, db, ws):
    _check(load().mcg_colsum_from_partials(Cn, _p(part), n_slots, slot_stride, _p(db), _p(ws), _stream()), "mcg_colsum_from_partials")


def randn_rowquad(out, Cn, sigma, seed, stream_id):
    """out [M][Cn] in the element order of the fused first-layer epilogue"""
    _check(load().mcg_randn_rowquad(out.numel() // Cn, Cn, sigma, seed, stream_id, _p(_dense(out)), _stream()), "mcg_randn_rowquad")


def colsum_acc(M, Cn, g, db, ws):
    _check(load().mcg_colsum_acc(M, Cn, _p(_dense(g)), _p(db), _p(ws), _stream()), "mcg_colsum_acc")


def pack_clip(N, Cn, Cp, T, HW, x, out, addend=None, sigma=0.0, seed=0, stream_id=0, stride_n=None, stride_c=None):
    """x: reference-layout (N,C,T,H,W) tensor, or a frame view of one with explicit strides."""
    if stride_n is None and stride_c is None:
        _dense(x)                                          # (default strides are those of a dense tensor: a permuted view would be misread)
    stride_n = Cn * T * HW if stride_n is None else stride_n
    stride_c = T * HW if stride_c is None else stride_c
    _check(load().mcg_pack_clip(N, Cn, Cp, T, HW, _p(x), stride_n, stride_c, _p(_dense(addend)), sigma, seed, stream_id,
                                _p(_dense(out)), _stream()), "mcg_pack_clip")


def pack_clip_u8(N, Cn, Cp, T, HW, x, out, addend=None, sigma=0.0, seed=0, stream_id=0, stride_n=None, stride_t=None):
    """x: the loader's uint8 clips (N,T,H,W,C), or frame x[:, t] of them (T = 1, the clip's stride_n): -> (x - 128) / 128 in the
    device layout [N][T][HW][Cp] (+ noise as pack_clip)."""
    if stride_n is None and stride_t is None:
        _dense(x)
    stride_t = HW * Cn if stride_t is None else stride_t
    stride_n = T * stride_t if stride_n is None else stride_n
    _check(load().mcg_pack_clip_u8(N, Cn, Cp, T, HW, _p(x, torch.uint8), stride_n, stride_t, _p(_dense(addend)), sigma, seed, stream_id,
                                   _p(_dense(out)), _stream()), "mcg_pack_clip_u8")


def concat_label_planes(x, c, dl, labels, out):
    """x [N][..][Cp] -> out [N][..][Cq]: the first c channels, dl label planes (+1 / -1), zero padding (model/updater.py:65-76);
    dl = 0: a channel slice into another row width."""
    N = x.shape[0]
    P = x[0].numel() // x.shape[-1]
    assert out.shape[:-1] == x.shape[:-1]
    _check(load().mcg_concat_label_planes(N, P, c, x.shape[-1], dl, out.shape[-1], _p(_dense(x)), _p(labels, torch.int32) if dl else None,
                                          _p(_dense(out)), _stream()), "mcg_concat_label_planes")
    return out


def unpack_clip(N, Cn, Cp, T, HW, inp, x):
    _check(load().mcg_unpack_clip(N, Cn, Cp, T, HW, _p(_dense(inp)), _p(_dense(x)), _stream()), "mcg_unpack_clip")


def tanh_bwd_to_frames(N, T, frame_elems, g_clip, x_clip, g_frames):
    _check(load().mcg_tanh_bwd_to_frames(N, T, frame_elems, _p(_dense(g_clip)), _p(_dense(x_clip)), _p(_dense(g_frames)),
                                         _stream()), "mcg_tanh_bwd_to_frames")


def gru_seq_fwd(N, T, dz, dl, dc, params, h0, e, labels, zc, z, saved):
    _check(load().mcg_gru_seq_fwd(N, T, dz, dl, dc, _p(params), _p(_dense(h0)), _p(_dense(e)), _p(labels, torch.int32),
                                  _p(_dense(zc)), _p(_dense(z)), _p(_dense(saved)), _stream()), "mcg_gru_seq_fwd")


def gru_seq_bwd(N, T, dz, dl, dc, params, e, labels, saved, gz, dparams):
    _check(load().mcg_gru_seq_bwd(N, T, dz, dl, dc, _p(params), _p(_dense(e)), _p(labels, torch.int32), _p(_dense(saved)),
                                  _p(_dense(gz)), _p(dparams), _stream()), "mcg_gru_seq_bwd")


def loss_dis(N, Cn, y_real, y_fake, t_real, t_fake, with_ce, loss_out, g_real, g_fake):
    _check(load().mcg_loss_dis(N, Cn, _p(_dense(y_real)), _p(_dense(y_fake)), _p(t_real, torch.int32), _p(t_fake, torch.int32),
                               int(with_ce), _p(loss_out), _p(_dense(g_real)), _p(_dense(g_fake)), _stream()), "mcg_loss_dis")


def loss_gen(N, Cn, y_i, y_v, t_fake, with_ce, loss_out, g_i, g_v):
    _check(load().mcg_loss_gen(N, Cn, _p(_dense(y_i)), _p(_dense(y_v)), _p(t_fake, torch.int32), int(with_ce), _p(loss_out),
                               _p(_dense(g_i)), _p(_dense(g_v)), _stream()), "mcg_loss_gen")


def adam_wd(p, g, m, v, lr_t, beta1, beta2, eps, wd, grad_scale=1.0, p16=None):
    """p16: optional bf16 buffer of p's size that receives a copy of the updated parameters"""
    _check(load().mcg_adam_wd(p.numel(), _p(_dense(p)), _p(_dense(g)), _p(_dense(m)), _p(_dense(v)), lr_t, beta1, beta2, eps, wd,
                              grad_scale, _p(_dense(p16), torch.bfloat16), _stream()), "mcg_adam_wd")


def randint(out, modulus, seed, stream_id):
    """out (int32)[i] = Philox word i of the stream, modulo `modulus` (oracle.philox.randint states the same draw)"""
    _check(load().mcg_randint(out.numel(), int(modulus), seed, stream_id, _p(_dense(out), torch.int32), _stream()), "mcg_randint")


class SplitSeg(C.Structure):                                       # mcg_split_seg
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64), ("run", C.c_int64)]


split_multi_launches = 0                                       # (tests assert that the one-launch refresh really ran)


def split_planes_multi(items):
    """items: [(fp32 source tensor, run, bf16 destination tensor of 4x the elements), ...] -- mcg_split_planes of every item in ONE
    launch (an 'f32x3' network's filters right after its Adam update).  The split operands count as GEMM time in bench.py's
    roofline leg, as split_planes' do."""
    global split_multi_launches
    split_multi_launches += 1
    segs = (SplitSeg * len(items))()
    for q, (src, run, dst) in zip(segs, items):
        assert src.is_contiguous() and dst.is_contiguous() and dst.dtype == torch.bfloat16 and dst.numel() == 4 * src.numel()
        q.src, q.dst, q.n, q.run = src.data_ptr(), dst.data_ptr(), src.numel(), int(run)
    e0 = None
    if _timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load().mcg_split_planes_multi(len(items), C.cast(segs, C.c_void_p), _stream()), "mcg_split_planes_multi")
    if e0 is not None:
        e1.record()
        _timing.setdefault(_tag + ".split", []).append((e0, e1))


def split_planes(src, run=16, out=None):
    """fp32 tensor -> its MCG_PREC_SPLIT form (include/mocogan_hip.h: mcg_split_planes): per run of `run` values four runs of
    bf16 (hi, mid, lo, padding).  run = 16: channels-last tensors (last dimension a multiple of 16) -> last dimension x 4."""
    src = _dense(src)
    n = src.numel()
    if out is None:
        out = torch.empty(src.shape[:-1] + (4 * src.shape[-1],), device=src.device, dtype=torch.bfloat16)
    assert out.numel() == 4 * n and out.dtype == torch.bfloat16
    e0 = None
    if _timing is not None:                                      # (bench.py's roofline leg charges the pass to the network's GEMMs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load().mcg_split_planes(n, int(run), _p(src), _p(out, torch.bfloat16), _stream()), "mcg_split_planes")
    if e0 is not None:
        e1.record()
        _timing.setdefault(_tag + ".split", []).append((e0, e1))
    return out


def randn(out, sigma, seed, stream_id):
    _check(load().mcg_randn(out.numel(), sigma, seed, stream_id, _p(_dense(out)), _stream()), "mcg_randn")
