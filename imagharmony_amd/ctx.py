"""Execution context: maps torch tensors (device memory + streams only -- plumbing) onto the
C ABI of libimh_hip.so, either eagerly (one ctypes call per op) or by RECORDING the calls into a
C++ plan that is later replayed / captured into a hipGraph (one UNet forward is ~1000 launches;
issuing them from Python would cost more than the GPU time).

Nothing here computes with torch: tensors are allocated, viewed and handed over as raw pointers.
"""
import ctypes as C
import json
import os

import torch

from . import lib as L

_DT = {torch.bfloat16: L.IMH_DT_BF16, torch.float16: L.IMH_DT_F16}
_TUNING_PATH = os.environ.get("IMH_TUNING_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning.json")   # override: A/B runs


_TUNING_CACHE = None


def _load_tuning():
    """parsed once per process (every eager processor call builds a Ctx)"""
    global _TUNING_CACHE
    if _TUNING_CACHE is None:
        _TUNING_CACHE = _read_tuning()
    return _TUNING_CACHE


def _read_tuning():
    try:
        with open(_TUNING_PATH) as f:
            raw = json.load(f)
        extra = os.environ.get("IMH_TUNING_OVERRIDE")      # A/B aid: a JSON object of entries laid over tuning.json, e.g. '{"2048,10240,1280,0,1": [23256, 160, 1]}'
        if extra:
            raw.update(json.loads(extra))
        return {tuple(int(v) for v in k.split(",")): tuple(cfg) for k, cfg in raw.items()}
    except (OSError, ValueError):
        return {}


class GnStats:
    """GroupNorm partials of ONE tensor (csrc/imh_lnstats.h gn_emit / norm.hip gn_stats_kernel): t [B, nblk, C / sub, 2] fp32 =
    (sum, M2) per (sample, pixel block, sub-run of `sub` channels); npart = elements per partial (0: the ragged blocks of the
    stand-alone statistics kernel).  Travels with the tensor from the launch that wrote it to every GroupNorm that reads it
    (directly or through a channel concat)."""
    __slots__ = ("t", "nblk", "sub", "npart", "C")

    def __init__(self, t, nblk, sub, npart, C_):
        self.t, self.nblk, self.sub, self.npart, self.C = t, int(nblk), int(sub), int(npart), int(C_)


class GnSpec:
    """A GroupNorm as its consumer's prologue sees it (round 5): the statistics of the input (one GnStats, or two for a channel concat),
    gamma / beta, groups, eps -- the consuming launch builds the (scale, shift) table of its sample itself (imh_gemm_args.gn_part /
    IMH_GN_TABLE_APPLY) instead of reading one written by a table launch."""
    def __init__(self, stats, gamma, beta, groups, eps):
        self.srcs = list(stats) if isinstance(stats, (list, tuple)) else [stats]
        self.gamma, self.beta, self.groups, self.eps = gamma, beta, int(groups), float(eps)
        self.C = sum(g.C for g in self.srcs)

    def check(self, B, Cc, descr):
        if not 1 <= len(self.srcs) <= 2 or self.C != Cc:
            raise L.ImhError(f"{descr}: statistics of {self.C} channels from {len(self.srcs)} source(s) do not fit C={Cc}")
        cpg = Cc // self.groups
        for g in self.srcs:
            if tuple(g.t.shape) != (B, g.nblk, g.C // g.sub, 2) or g.t.dtype != torch.float32 or cpg % g.sub or self.srcs[0].C % g.sub:
                raise L.ImhError(f"{descr}: statistics {tuple(g.t.shape)} (sub {g.sub}) do not fit C={Cc}, groups={self.groups}")

    def tensors(self):
        return tuple(g.t for g in self.srcs) + (self.gamma, self.beta)


class Ctx:
    def __init__(self, device, dtype=torch.bfloat16, record=False, dry=False):
        """dry=True (record only): tensors may live on the CPU; the plan can be inspected (op list,
        FLOP / byte accounting) but never run -- used by the host-logic tests, not a compute path."""
        self.lib = L.load()
        self.device = torch.device(device)
        self.dry = bool(dry and record)
        if self.device.type != "cuda" and not self.dry:
            raise L.ImhError("imagharmony_amd needs a ROCm device (there is no CPU path)")
        if dtype not in _DT:
            raise L.ImhError(f"unsupported compute dtype {dtype}")
        self.dtype = dtype
        self.dt = _DT[dtype]
        self.record = record
        self.plan = self.lib.imh_plan_create() if record else None
        self.keep = []          # tensors referenced by recorded ops
        self.tag = 0
        self.tags = []          # per recorded op: (tag, kind, descr, flops, bytes, shape, epilogue/config dict)
        self._pool = {}
        self._bases = {}        # storage ptr -> pool-owned base tensor
        self._live = set()
        self._ws = None
        self.tuning = _load_tuning()
        # XCD cell shape every GEMM / implicit-GEMM launch of this context asks for (imh_gemm_args.xcd: 0 = the byte-count model,
        # 2 .. 5 = 8 x 1 / 4 x 2 / 2 x 4 / 1 x 8 cells over M x N; + 10: the GEGLU launches (N = 8 C) keep the model's N-major
        # choice); placement only -- DenoiseEngine measures which one this box prefers
        self.xcd_cells = 0
        self.captured = False
        self._ops = []          # recorded (kind, ctypes args, cold tensors) for the weight-prefetch pass
        self._pf_done = False

    # ------------------------------------------------------------------ memory
    def new(self, *shape, dtype=None):
        dtype = dtype or self.dtype
        n = 1
        for s in shape:
            n *= int(s)
        nb = n * torch.empty((), dtype=dtype).element_size()
        key = (nb + 255) // 256 * 256
        lst = self._pool.get(key)
        if lst:
            base = lst.pop()
        else:
            base = torch.empty(key, dtype=torch.uint8, device=self.device)
            self._bases[base.data_ptr()] = base
            if self.record:
                self.keep.append(base)
        self._live.add(base.data_ptr())
        return base[:nb].view(dtype).view(*shape)

    def free(self, t):
        """Return an activation buffer (or any view of it) to the pool.  Only legal once every op
        that reads it has been emitted (stream order makes later reuse safe)."""
        ptr = t.untyped_storage().data_ptr()
        base = self._bases.get(ptr)
        if base is None:
            return                      # not pool-owned (weights, caller tensors)
        if ptr not in self._live:
            raise L.ImhError("Ctx.free: buffer freed twice")
        self._live.discard(ptr)
        self._pool.setdefault(base.numel(), []).append(base)

    def zeros(self, *shape, dtype=None):
        t = torch.zeros(*shape, dtype=dtype or self.dtype, device=self.device)
        if self.record:
            self.keep.append(t)
        return t

    def workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            # (a recording context keeps one generous slab for the whole plan; an eager context -- one per conditioning-module call --
            # takes what the launch needs: 64 MB per call was a device allocation per Resampler pass)
            floor = (64 << 20) if self.record else (1 << 20)
            self._ws = torch.empty(max((int(nbytes) + (1 << 20) - 1) >> 20 << 20, floor), dtype=torch.uint8, device=self.device)
            if self.record:
                self.keep.append(self._ws)
        return self._ws

    def stream(self):
        if self.dry:
            raise L.ImhError("a dry-run context cannot execute (no CPU path)")
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ emit
    def _emit(self, kind, args, ew_op=0, descr="", flops=0.0, nbytes=0.0, keep=(), shape=None, epi=None):
        if self.record:
            rc = self.lib.imh_plan_add(self.plan, kind, C.byref(args), ew_op, self.tag)
            if rc < 0:
                L.check(rc, "imh_plan_add")
            self.keep.extend(k for k in keep if k is not None)
            self.tags.append((self.tag, kind, descr, flops, nbytes, shape, epi))
            self._ops.append((kind, args, self._cold(keep) if kind in (L.OP_GEMM, L.OP_XATTN) else []))
            return
        s = self.stream()
        if kind == L.OP_GEMM:
            rc = self.lib.imh_gemm(C.byref(args), s)
        elif kind == L.OP_ATTN:
            rc = self.lib.imh_attention(C.byref(args), s)
        elif kind == L.OP_GROUPNORM:
            rc = self.lib.imh_groupnorm(C.byref(args), s)
        elif kind == L.OP_LAYERNORM:
            rc = self.lib.imh_layernorm(C.byref(args), s)
        elif kind == L.OP_ATTN_SMALL:
            rc = self.lib.imh_attention_small(C.byref(args), s)
        elif kind == L.OP_XATTN:
            rc = self.lib.imh_cross_attention(C.byref(args), s)
        else:
            rc = self.lib.imh_elementwise(ew_op, C.byref(args), s)
        L.check(rc, descr or f"op kind {kind}")

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def _chk(self, t, name, dtype=None):
        if t is None:
            return
        if t.device != self.device and not (t.device.type == "cuda" and t.device.index == (self.device.index or 0)):
            raise L.ImhError(f"{name}: tensor on {t.device}, context on {self.device}")
        if t.dtype != (dtype or self.dtype):
            raise L.ImhError(f"{name}: dtype {t.dtype}, expected {dtype or self.dtype}")

    # ------------------------------------------------------------------ GEMM / conv
    # variant codes (include/imh.h): what each family can do, so that a tuning-table entry (keyed by shape only) is never
    # handed a launch it rejects
    _WS = (1464, 2464, 24128, 23256, 22128)   # wave-specialised (gemm_ring.hip); 22128: two workgroups per CU
    _W16 = (26256,)                           # sixteen-wave 256 x 320 (gemm_w16.hip): row-form folded LayerNorm (+ GEGLU) launches only
    _PP = (8256, 9128, 9256)                  # ping-pong (gemm_pp.hip)
    _HALO = (7128, 7564, 7328, 7428, 7256, 7356)   # LDS-halo conv3x3, stride 1 (7328 / 7428: weight rings; 7256 / 7356: 16 x 16 patch)

    @classmethod
    def _variant_ok(cls, bm, sp, flags, conv, stride, ln_pre):
        plain = bm <= 128
        if bm in cls._W16:
            return not conv and sp == 1 and bool(flags & L.GF_LN_ROW) and not flags & ~(L.GF_LN_ROW | L.GF_GEGLU) and bool(ln_pre)
        if bm in cls._HALO:
            return bool(conv) and stride == 1 and not flags & (L.GF_LN_ROW | L.GF_LN_COL | L.GF_VT_PERM)
        if flags & (L.GF_VT_PERM | L.GF_LN_COL):
            return plain and (sp == 1 or not flags & L.GF_LN_COL)
        if flags & L.GF_LN_ROW:
            return sp == 1 and not conv and (plain or bm in cls._WS)
        if bm in cls._PP and conv:
            return False
        return True

    def _xcd_for(self, N, flags, conv):
        """imh_gemm_args.xcd of one launch under this context's policy (xcd_cells): 0 / 2..5 = the same request for every launch;
        1x = x for every launch except the GEGLU ones (model).  (A policy that asked for whole-row cells only on the Linear launches
        with small weights -- N <= 1280 -- measured no different from the model on two boxes where 4 x 2 / 8 x 1 EVERYWHERE were 0.2-1.0
        ms ahead: profiles/r04_forward_ab_xcd_policies_box*.json; dropped.)"""
        pol, cells = divmod(self.xcd_cells, 10)
        if pol == 1 and flags & L.GF_GEGLU:
            return 0
        return cells

    def _config(self, M, N, K, conv, flags, stride=1, ln_pre=False, up=0):
        """tile variant of a launch: the tuning table's entry for the shape if that variant implements the launch's flags
        (folded LayerNorm in either form, V^T permutation, conv stride), else the built-in heuristic -- in ONE place"""
        key = (M, N, K, int(conv))
        # a fifth key field: 1 = the entry for launches whose LayerNorm statistics are precomputed (other variants apply);
        # 2 = the entry for a stride-2 conv whose (M, N, K) coincides with a stride-1 conv of another resolution / batch;
        # 3 = the entry for a conv with fused x2 upsampling (round 6: the 64^2 -> 128^2 upsampler of UNet batch 2 shares (M, N, K) with the
        # 64^2 ResBlock convs of UNet batch 8, which want the LDS-halo form that fuses their GroupNorm)
        cfg = (self.tuning.get(key + (1,)) if ln_pre else self.tuning.get(key + (2,)) if conv and stride == 2
               else self.tuning.get(key + (3,)) if conv and up else None) or self.tuning.get(key)
        if cfg is not None and self._variant_ok(cfg[0], cfg[2], flags, conv, stride, ln_pre):
            return tuple(cfg)
        bm, bn, sp = C.c_int(), C.c_int(), C.c_int()
        self.lib.imh_gemm_pick_config(M, N, K, C.byref(bm), C.byref(bn), C.byref(sp))
        bm, bn, sp = bm.value, bn.value, sp.value
        if flags & (L.GF_LN_ROW | L.GF_LN_COL) and (sp > 1 or cfg is not None):
            # the statistics need the whole K range in one workgroup; a rejected table entry means a small-tile problem
            bm, bn, sp = (128, 128, 1) if M * N >= 8192 * 640 else (64, 64, 1)
        return bm, bn, sp

    def gemm(self, x, w, out=None, bias=None, residual=None, rowadd=None, rows_per_batch=0, ldra=0, flags=0,
             M=None, N=None, K=None, ldx=None, ldw=None, ldy=None, ldr=None, cfg=None, descr="gemm", out_dtype=None,
             ln=None, stats_out=False, gn_out=None, x2=None, yt=None, _args_only=False):
        """y[M, N] = epilogue(x[M, K] @ w[N, K]^T).  x / w: 2-D, last dim contiguous.
        ln = (s, c, eps[, stats]) with GF_LN_ROW / GF_LN_COL in flags: LayerNorm of the token operand folded into the GEMM
        (w pre-scaled by gamma); stats = (tensor [tokens, slots, 2] fp32, slots) are the token rows' statistics as handed
        over by the launch that wrote them (csrc/imh_lnstats.h); None -> a row-statistics launch over the token rows
        supplies them (the kernels have no in-loop E[x^2] - mean^2 form).
        stats_out=True: also return the row statistics of y for a LayerNorm-folding consumer -> (y, (tensor, slots)); they
        come from the GEMM's own epilogue when the chosen variant has one, else from a row-statistics launch over y.
        gn_out=hw (rows per sample; a (groups, hw) pair is accepted, the groups are the consumer's business): y is a GroupNorm
        input -> (y, gn) with gn = GnStats from the epilogue, or None when the chosen variant has no such epilogue (the consumer
        then runs gn_stats over y).
        yt = (Yt [N - col0, ldyt], col0): output columns >= col0 are stored transposed in the V^T layout into Yt instead of y (y
        then holds columns [0, col0)); wave-specialised bn = 160 variants, flags == GF_LN_ROW only (the one-launch [Q|K|V]).
        x2: the token operand is the column concat [x | x2] (K = x.shape[1] + x2.shape[1]; the up blocks' conv_shortcut over
        torch.cat([hidden, skip], 1)) read from its two producers -- plain 64 / 128 tiles and the wave-specialised variants."""
        self._chk(x, descr + ".x"); self._chk(w, descr + ".w")
        M = M if M is not None else x.shape[0]
        K = K if K is not None else x.shape[1] + (x2.shape[1] if x2 is not None else 0)
        N = N if N is not None else w.shape[0]
        if x.stride(-1) != 1 or w.stride(-1) != 1:
            raise L.ImhError(f"{descr}: operands must be contiguous in K")
        if x2 is not None:
            self._chk(x2, descr + ".x2")
            if x2.shape[0] != x.shape[0] or x2.stride(-1) != 1 or x.shape[1] % 64 or x2.shape[1] % 64 or x.shape[1] + x2.shape[1] != K \
                    or x2.stride(0) != x2.shape[1] or (flags & (L.GF_LN_ROW | L.GF_LN_COL | L.GF_VT_PERM)):
                raise L.ImhError(f"{descr}: x2 must be a dense [M, K2] block with K1, K2 multiples of 64 and K1 + K2 == K (no folded LayerNorm)")
        if isinstance(gn_out, tuple):
            gn_out = gn_out[1]
        n_out = N // 2 if flags & L.GF_GEGLU else N
        if yt is not None:
            n_out = int(yt[1])
        if out is None:
            out = self.new(M, n_out, dtype=torch.float32 if flags & L.GF_OUT_F32 else None)
        ln_stats = ln[3] if ln is not None and len(ln) > 3 else None
        own_stats = False
        if ln is not None and ln_stats is None and flags & (L.GF_LN_ROW | L.GF_LN_COL):
            if _args_only:
                raise L.ImhError(f"{descr}: a folded-LayerNorm problem of gemm_dual needs its row statistics (gemm_dual supplies them)")
            tok = x if flags & L.GF_LN_ROW else w
            ln_stats = self.row_stats(tok[:M if flags & L.GF_LN_ROW else N, :K], descr=descr + ".ln_row_stats")
            own_stats = True
        if stats_out and flags & L.GF_OUT_F32:
            raise L.ImhError(f"{descr}: stats_out describes rows stored in the compute dtype; an fp32 output (GF_OUT_F32) has no such statistics")
        if stats_out and gn_out is not None:
            raise L.ImhError(f"{descr}: stats_out and gn_out are mutually exclusive (one consumer norm per output)")
        bm, bn, sp = cfg or self._config(M, N, K, 0, flags, ln_pre=ln_stats is not None)
        if x2 is not None and not (bm <= 128 or bm in self._WS):
            if cfg is not None:
                raise L.ImhError(f"{descr}: variant {bm} does not read a two-source token operand")
            bm, bn, sp = (128, 128, 1) if M * N >= 8192 * 640 else (64, 64, 1)
        if _args_only:
            sp = 1
        a = L.GemmArgs()
        a.X, a.W, a.Y = x.data_ptr(), w.data_ptr(), out.data_ptr()
        if x2 is not None:
            a.X2, a.Cin1 = x2.data_ptr(), x.shape[1]
        if yt is not None:
            self._chk(yt[0], descr + ".yt")
            if yt[0].dim() != 2 or yt[0].stride(1) != 1 or yt[0].shape[0] != N - int(yt[1]) or yt[0].shape[1] < M:
                raise L.ImhError(f"{descr}: yt {tuple(yt[0].shape)} does not fit [{N - int(yt[1])}, >= {M}]")
            a.Yt, a.yt_col0, a.ldyt = yt[0].data_ptr(), int(yt[1]), yt[0].stride(0)
        a.bias, a.rowadd, a.residual = self._p(bias), self._p(rowadd), self._p(residual)
        keep_ln = ()
        if ln is not None:           # (s, c fp32, eps[, stats]); the caller sets GF_LN_ROW / GF_LN_COL in flags
            a.ln_s, a.ln_c, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
            keep_ln = (ln[0], ln[1])
            if ln_stats is not None:
                a.ln_stats, a.ln_slots = ln_stats[0].data_ptr(), int(ln_stats[1])
                keep_ln += (ln_stats[0],)
        a.M, a.N, a.K = M, N, K
        a.ldx = ldx if ldx is not None else x.stride(0)
        a.ldw = ldw if ldw is not None else w.stride(0)
        a.ldy = ldy if ldy is not None else out.stride(0)
        a.ldr = (ldr if ldr is not None else residual.stride(0)) if residual is not None else 0
        a.ldra = ldra
        a.rows_per_batch = rows_per_batch
        a.splits, a.flags, a.dtype, a.conv, a.bm, a.bn = sp, flags, self.dt, 0, bm, bn
        a.xcd = self._xcd_for(N, flags, 0)
        if sp > 1:
            a.partial = self.workspace(self.lib.imh_gemm_workspace_bytes(M, N, sp)).data_ptr()
        st = None
        if stats_out and not _args_only:
            wd = self.lib.imh_gemm_stats_slot_width(bm, bn)
            if wd > 0 and N % wd == 0 and sp == 1 and not flags & (L.GF_GEGLU | L.GF_VT_PERM | L.GF_OUT_F32):
                st = (self.new(M, N // wd, 2, dtype=torch.float32), N // wd)
                a.ln_stats_out, a.ln_slots_out = st[0].data_ptr(), st[1]
        gn = self._gn_epilogue(a, gn_out) if gn_out is not None and not _args_only else None
        es = x.element_size()
        if _args_only:
            return a, out, 2.0 * M * N * K, es * (M * K + N * K + M * n_out), (x, w, out, bias, rowadd, residual, x2) + keep_ln
        self._emit(L.OP_GEMM, a, descr=descr, flops=2.0 * M * N * K, nbytes=es * (M * K + N * K + M * n_out),
                   keep=(x, w, out, bias, rowadd, residual, x2) + keep_ln + ((st[0],) if st else ()) + ((gn.t,) if gn else ())
                   + ((yt[0],) if yt is not None else ()),
                   shape=(M, N, K, 0, None),
                   epi=dict(flags=flags, bias=bias is not None, residual=residual is not None, rowadd=rowadd is not None,
                            rows_per_batch=rows_per_batch, cfg=(bm, bn, sp), ln_pre=ln_stats is not None,
                            ln_slots=int(ln_stats[1]) if ln_stats is not None else 0, stats_out=st is not None,
                            gn_out=(gn.nblk, gn_out) if gn else None, x2=x.shape[1] if x2 is not None else 0,
                            yt=int(yt[1]) if yt is not None else 0))
        if own_stats:
            self.free(ln_stats[0])
        if stats_out:
            if st is None:
                st = self.row_stats(out.view(M, n_out) if out.dim() != 2 else out, descr=descr + ".row_stats")
            return out, st
        if gn_out is not None:
            return out, gn
        return out

    def _gn_epilogue(self, a, hw):
        """GroupNorm partials from the launch's epilogue (imh_gemm_args.gn_out) if its variant and shape have one: fills the
        fields of a and returns the GnStats of the output ([B, blocks, N / 10, 2] fp32), else None"""
        rows = self.lib.imh_gemm_gn_block_rows(a.bm, a.bn)
        halo = a.bm in self._HALO
        if (rows <= 0 or a.splits != 1 or a.N % 10 or hw % rows or a.M % hw
                or a.N % (a.bn if halo else 80) or a.flags & ~(L.GF_ACT_SILU | L.GF_ACT_GELU)
                or (halo and (a.Ho % (rows // 4) or a.Wo % 16))):
            return None
        nblk = hw // rows
        t = self.new(a.M // hw, nblk, a.N // 10, 2, dtype=torch.float32)
        a.gn_out, a.gn_nblk, a.gn_hw = t.data_ptr(), nblk, hw
        return GnStats(t, nblk, 10, rows * 10, a.N)

    def row_stats(self, x, descr="row_stats"):
        """LayerNorm statistics of token rows x [rows, C] in the hand-over format (one slot per row): the stand-alone
        producer for rows whose writing GEMM variant has no statistics epilogue"""
        rows, Cc = x.shape
        st = self.new(rows, 1, 2, dtype=torch.float32)
        self.ew(L.EW_ROW_STATS, st, a=x, n=rows, i=(Cc, x.stride(0), 0, 0, 0, 0), descr=descr,
                nbytes=float(x.numel() * x.element_size()))
        return st, 1

    def gemm_dual(self, g1, g2, cfg=(128, 64), descr="gemm_dual"):
        """Two independent GEMMs (dicts of gemm() keyword arguments incl. x, w) in one launch.  The folded-LayerNorm pair
        (g1 row form on x, g2 column form with the same token rows as its w) shares one statistics tensor; when the caller has
        none, a row-statistics launch over the token rows supplies it."""
        for g in (g1, g2):
            if g.get("x2") is not None or g.get("yt") is not None or g.get("gn_out") is not None or g.get("stats_out"):
                raise L.ImhError(f"{descr}: x2 / yt / gn_out / stats_out are single-problem (Ctx.gemm) features")
        own = None
        l1, l2 = g1.get("ln"), g2.get("ln")
        if l1 is not None and (len(l1) < 4 or l1[3] is None):
            own = self.row_stats(g1["x"], descr=descr + ".ln_row_stats")
            g1 = dict(g1, ln=tuple(l1[:3]) + (own,))
            if l2 is not None and (len(l2) < 4 or l2[3] is None):
                g2 = dict(g2, ln=tuple(l2[:3]) + (own,))
        a1, o1, f1, b1, k1 = self.gemm(_args_only=True, cfg=(cfg[0], cfg[1], 1), **g1)
        a2, o2, f2, b2, k2 = self.gemm(_args_only=True, cfg=(cfg[0], cfg[1], 1), **g2)
        pair = (L.GemmArgs * 2)(a1, a2)
        if self.record:
            rc = self.lib.imh_plan_add(self.plan, L.OP_GEMM_DUAL, C.cast(pair, C.c_void_p), 0, self.tag)
            if rc < 0:
                L.check(rc, "imh_plan_add")
            self.keep.extend(t for t in k1 + k2 if t is not None)
            self.tags.append((self.tag, L.OP_GEMM, descr, f1 + f2, b1 + b2, None,
                              dict(dual=((a1.M, a1.N, a1.K, a1.flags), (a2.M, a2.N, a2.K, a2.flags)), cfg=tuple(cfg),
                                   ln_pre=bool(a1.ln_stats), ln_slots=int(a1.ln_slots))))
            self._ops.append((L.OP_GEMM_DUAL, pair, self._cold(k1[:2] + k2[:2])))
        else:
            L.check(self.lib.imh_gemm_dual(C.byref(pair[0]), C.byref(pair[1]), self.stream()), descr)
        if own is not None:
            self.free(own[0])
        return o1, o2

    def conv_fuses_gn(self, M, N, K, stride=1, up=0, cfg=None):
        """True when the conv3x3 launch of this shape runs on the LDS-halo kernel, which takes the GroupNorm (+ SiLU) of its input
        (gn=...) and a two-source channel concat (x2=...) in its halo staging"""
        bm, bn, sp = cfg or self._config(M, N, K, 1, 0, stride=stride, up=up)
        if bm not in self._HALO or stride != 1 or up:
            return False
        ph = 4 if bm == 7564 else (16 if bm in (7256, 7356) else 8)
        S = (6 if ph == 16 else 9) if bn == 80 else (3 if bm in (7328, 7356) else (4 if bm == 7428 else 2))     # (x 80: two / three slots of three tap tiles each)
        lds = 2 * (((ph + 2) * 18 + 7) // 8) * 8 * 128 + S * bn * 128 + (K // 9) * 8
        return lds <= 160 * 1024

    def conv3x3(self, x, w, bias=None, stride=1, up=0, residual=None, rowadd=None, ldra=0, out=None, cfg=None,
                descr="conv3x3", gn_groups=0, gn=None, x2=None):
        """x: NHWC [B, H, W, Cin]; w: packed [Cout, 9*Cin]; returns NHWC [B, Ho, Wo, Cout]; with gn_groups > 0 (the output is
        a GroupNorm input) -> (y, GnStats or None) as gemm(gn_out=...).
        gn = (table [B, Cin, 2] fp32 from gn_table(), silu) or (GnSpec, silu): the input's GroupNorm (+ SiLU) is applied inside the
        kernel's halo staging (diffusers ResnetBlock2D: norm -> nonlinearity -> conv in one launch) -- with a GnSpec the kernel also builds
        the table itself from the producers' partials (no table launch); x2: the input is the channel concat [x | x2].  Both need the
        LDS-halo variant (conv_fuses_gn)."""
        self._chk(x, descr + ".x"); self._chk(w, descr + ".w")
        B, H, W, C1 = x.shape
        Cin = C1 + (x2.shape[-1] if x2 is not None else 0)
        Cout = w.shape[0]
        Hv, Wv = H << up, W << up
        Ho, Wo = (Hv - 1) // stride + 1, (Wv - 1) // stride + 1
        M, N, K = B * Ho * Wo, Cout, 9 * Cin
        if not x.is_contiguous() or not w.is_contiguous() or w.shape[1] != K:
            raise L.ImhError(f"{descr}: x must be contiguous NHWC and w packed [Cout, 9*Cin]")
        if x2 is not None:
            self._chk(x2, descr + ".x2")
            if tuple(x2.shape[:3]) != (B, H, W) or not x2.is_contiguous() or C1 % 64 or x2.shape[-1] % 64:
                raise L.ImhError(f"{descr}: x2 must be contiguous NHWC over the same pixels, both channel counts multiples of 64")
        if out is None:
            out = self.new(B, Ho, Wo, Cout)
        # (the table is keyed by (M, N, K): a stride-2 conv can share its key with a stride-1 conv of another resolution /
        # batch; the LDS-halo kernel is stride-1 only -> _config falls back to the heuristic tile for that one)
        bm, bn, sp = cfg or self._config(M, N, K, 1, 0, stride=stride, up=up)
        if (gn is not None or x2 is not None) and not self.conv_fuses_gn(M, N, K, stride, up, cfg=(bm, bn, sp)):
            raise L.ImhError(f"{descr}: the fused GroupNorm front end / two-source input need the LDS-halo conv3x3 (variant {bm} x {bn}, "
                             f"stride {stride}, up {up}); apply the GroupNorm / concat as passes for this launch (Ctx.conv_fuses_gn)")
        a = L.GemmArgs()
        a.X, a.W, a.Y = x.data_ptr(), w.data_ptr(), out.data_ptr()
        a.bias, a.rowadd, a.residual = self._p(bias), self._p(rowadd), self._p(residual)
        a.M, a.N, a.K = M, N, K
        a.ldx, a.ldw, a.ldy = Cin, K, Cout
        a.ldr = Cout if residual is not None else 0
        a.ldra = ldra
        a.rows_per_batch = Ho * Wo
        a.splits, a.flags, a.dtype, a.conv, a.bm, a.bn = sp, 0, self.dt, 1, bm, bn
        a.xcd = self._xcd_for(N, 0, 1)
        a.H, a.Wd, a.Cin, a.Ho, a.Wo, a.stride, a.up = H, W, Cin, Ho, Wo, stride, up
        if x2 is not None:
            a.X2, a.Cin1 = x2.data_ptr(), C1
        gn_keep = ()
        if gn is not None and isinstance(gn[0], GnSpec):
            sp_, silu = gn
            sp_.check(B, Cin, descr)
            s0 = sp_.srcs[0]
            a.gn_part, a.gn_pnblk, a.gn_psub, a.gn_pnpart, a.gn_pC1 = s0.t.data_ptr(), s0.nblk, s0.sub, s0.npart, s0.C
            if len(sp_.srcs) == 2:
                s1 = sp_.srcs[1]
                a.gn_part2, a.gn_pnblk2, a.gn_psub2, a.gn_pnpart2 = s1.t.data_ptr(), s1.nblk, s1.sub, s1.npart
            a.gn_gamma, a.gn_beta, a.gn_groups, a.gn_eps, a.gn_silu = self._p(sp_.gamma), self._p(sp_.beta), sp_.groups, sp_.eps, int(bool(silu))
            gn_keep = sp_.tensors()
        elif gn is not None:
            tab, silu = gn
            if tuple(tab.shape) != (B, Cin, 2) or tab.dtype != torch.float32 or not tab.is_contiguous():
                raise L.ImhError(f"{descr}: GroupNorm table {tuple(tab.shape)} does not fit [{B}, {Cin}, 2] fp32")
            a.gn_tab, a.gn_silu = tab.data_ptr(), int(bool(silu))
            gn_keep = (tab,)
        if sp > 1:
            a.partial = self.workspace(self.lib.imh_gemm_workspace_bytes(M, N, sp)).data_ptr()
        gs = self._gn_epilogue(a, Ho * Wo) if gn_groups else None
        es = x.element_size()
        self._emit(L.OP_GEMM, a, descr=descr, flops=2.0 * M * N * K,
                   nbytes=es * (B * H * W * Cin + N * K + M * N),
                   keep=(x, w, out, bias, rowadd, residual, x2) + ((gs.t,) if gs else ()) + gn_keep,
                   shape=(M, N, K, 1, (B, H, W, Cin, stride, up)),
                   epi=dict(flags=0, bias=bias is not None, residual=residual is not None, rowadd=rowadd is not None,
                            rows_per_batch=Ho * Wo, cfg=(bm, bn, sp), gn_out=(gs.nblk, Ho * Wo) if gs else None,
                            gn_in=None if gn is None else int(bool(gn[1])), x2=C1 if x2 is not None else 0))
        return (out, gs) if gn_groups else out

    # ------------------------------------------------------------------ attention
    def attention(self, q, k, vt, out, B, H, Lq, Lk, Lk_pad, ldq, ldk, ldvt, ldo, scale,
                  k2=None, vt2=None, Lk2=0, Lk2_pad=0, ldk2=0, ldvt2=0, scale2=0.0, scale2_tab=None, step=None,
                  descr="attention"):
        a = L.AttnArgs()
        a.Q, a.K, a.Vt, a.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
        a.K2, a.Vt2 = self._p(k2), self._p(vt2)
        a.B, a.H, a.Lq, a.Lk, a.Lk_pad, a.Lk2, a.Lk2_pad = B, H, Lq, Lk, Lk_pad, Lk2, Lk2_pad
        a.ldq, a.ldk, a.ldvt, a.ldk2, a.ldvt2, a.ldo = ldq, ldk, ldvt, ldk2, ldvt2, ldo
        a.scale, a.scale2, a.dtype = scale, scale2, self.dt
        a.scale2_tab, a.step = self._p(scale2_tab), self._p(step)
        es = q.element_size()
        fl = 4.0 * B * H * Lq * (Lk + Lk2) * 64
        by = es * (2 * B * Lq * H * 64 + 2 * B * (Lk + Lk2) * H * 64)
        self._emit(L.OP_ATTN, a, descr=descr, flops=fl, nbytes=by, keep=(q, k, vt, out, k2, vt2, scale2_tab, step))
        return out

    def cross_attention(self, x, wq, k, vt, out, B, H, Lq, Lk, Lk_pad, ldk, ldvt, scale, ln=None,
                        k2=None, vt2=None, Lk2=0, Lk2_pad=0, ldk2=0, ldvt2=0, scale2=0.0, scale2_tab=None, step=None,
                        descr="cross.fused"):
        """out[B*Lq, C] = attention(to_q(LN?(x)), K, V) (+ scale2 * attention(., K2, V2)) in one launch
        (csrc/xattn.hip).  x [B*Lq, C]; wq [C, C]; k / k2 caches with head dims in vt_perm16 order (GF_VT_PERM);
        ln = (s, c, eps[, stats]): x is un-normalised and wq pre-scaled by gamma; stats as in gemm() (None -> a row-statistics
        launch over x supplies them)."""
        self._chk(x, descr + ".x"); self._chk(wq, descr + ".wq")
        C_ = H * 64
        if x.shape[-1] != C_ or tuple(wq.shape) != (C_, C_) or x.stride(-1) != 1 or wq.stride(-1) != 1:
            raise L.ImhError(f"{descr}: x [.., {C_}] / wq [{C_}, {C_}] expected, got {tuple(x.shape)} / {tuple(wq.shape)}")
        a = L.XAttnArgs()
        a.X, a.Wq, a.K, a.Vt, a.O = x.data_ptr(), wq.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
        a.K2, a.Vt2 = self._p(k2), self._p(vt2)
        keep_st = ()
        own = False
        if ln is not None:
            a.ln_s, a.ln_c, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
            st_in = ln[3] if len(ln) > 3 and ln[3] is not None else None
            own = st_in is None
            if own:
                st_in = self.row_stats(x.view(-1, C_), descr=descr + ".ln_row_stats")
            a.ln_stats, a.ln_slots = st_in[0].data_ptr(), int(st_in[1])
            keep_st = (st_in[0],)
        a.B, a.H, a.Lq, a.C = B, H, Lq, C_
        a.Lk, a.Lk_pad, a.Lk2, a.Lk2_pad = Lk, Lk_pad, Lk2, Lk2_pad
        a.ldx, a.ldw, a.ldk, a.ldvt, a.ldk2, a.ldvt2, a.ldo = x.stride(0), wq.stride(0), ldk, ldvt, ldk2, ldvt2, out.stride(0)
        a.scale, a.scale2, a.dtype = scale, scale2, self.dt
        a.scale2_tab, a.step = self._p(scale2_tab), self._p(step)
        es = x.element_size()
        M = B * Lq
        fl = 2.0 * M * C_ * C_ + 4.0 * B * H * Lq * (Lk + Lk2) * 64
        by = es * (2 * M * C_ + C_ * C_ + 2 * B * (Lk + Lk2) * C_)
        self._emit(L.OP_XATTN, a, descr=descr, flops=fl, nbytes=by,
                   keep=(x, wq, k, vt, out, k2, vt2, scale2_tab, step) + tuple((ln or ())[:2]) + keep_st)
        if own:
            self.free(keep_st[0])
        return out

    # ------------------------------------------------------------------ fp32 (reference-precision) ops of the VAE decode tail (csrc/f32.hip)
    def _f32(self, op, a, descr):
        if self.record:
            raise L.ImhError(f"{descr}: the fp32 ops run eagerly (once per image); they are not plan ops")
        L.check(self.lib.imh_f32(op, C.byref(a), self.stream()), descr)

    def f32_gemm(self, x, w, bias=None, residual=None, out=None, M=None, N=None, K=None, ldx=None, ldw=None, descr="f32.gemm"):
        """y[M, N] = x[M, K] @ w[N, K]^T (+ bias) (+ residual), everything fp32 (v_mfma_f32_32x32x2_f32); K % 16 == 0"""
        for t, n in ((x, "x"), (w, "w"), (bias, "bias"), (residual, "residual"), (out, "out")):
            self._chk(t, f"{descr}.{n}", torch.float32)
        M = M if M is not None else x.shape[0]
        K = K if K is not None else x.shape[1]
        N = N if N is not None else w.shape[0]
        if out is None:
            out = self.new(M, N, dtype=torch.float32)
        a = L.F32Args()
        a.X, a.W, a.Y, a.bias, a.residual = x.data_ptr(), w.data_ptr(), out.data_ptr(), self._p(bias), self._p(residual)
        a.M, a.N, a.K = M, N, K
        a.ldx = ldx if ldx is not None else x.stride(0)
        a.ldw = ldw if ldw is not None else w.stride(0)
        a.ldy = out.stride(0)
        a.ldr = residual.stride(0) if residual is not None else 0
        self._f32(L.F32_GEMM, a, descr)
        return out

    def f32_conv3x3(self, x, w, bias=None, residual=None, up=0, descr="f32.conv3x3"):
        """x NHWC [B, H, W, Cin] fp32, w packed [Cout, 9 Cin] fp32 -> [B, H << up, W << up, Cout]; stride 1, padding 1, nearest x2 fused"""
        for t, n in ((x, "x"), (w, "w"), (bias, "bias"), (residual, "residual")):
            self._chk(t, f"{descr}.{n}", torch.float32)
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        Ho, Wo = H << up, W << up
        if not x.is_contiguous() or not w.is_contiguous() or w.shape[1] != 9 * Cin:
            raise L.ImhError(f"{descr}: x must be contiguous NHWC and w packed [Cout, 9*Cin]")
        out = self.new(B, Ho, Wo, Cout, dtype=torch.float32)
        a = L.F32Args()
        a.X, a.W, a.Y, a.bias, a.residual = x.data_ptr(), w.data_ptr(), out.data_ptr(), self._p(bias), self._p(residual)
        a.M, a.N, a.K = B * Ho * Wo, Cout, 9 * Cin
        a.ldx, a.ldw, a.ldy, a.ldr = Cin, 9 * Cin, Cout, (residual.stride(-2) if residual is not None else 0)
        a.conv, a.H, a.Wd, a.Cin, a.Ho, a.Wo, a.up = 1, H, W, Cin, Ho, Wo, up
        self._f32(L.F32_GEMM, a, descr)
        return out

    def f32_groupnorm(self, x, gamma, beta, groups, eps, silu=False, descr="f32.groupnorm"):
        """x [B, HW, C] fp32 -> GroupNorm(groups)(+ SiLU), fp32; statistics as shifted (sum, M2) partials merged in double"""
        for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta")):
            self._chk(t, f"{descr}.{n}", torch.float32)
        B, HW, Cc = x.shape
        nblk = max(1, (HW + 2047) // 2048)
        part = self.new(B, nblk, groups, 2, dtype=torch.float32)
        tab = self.new(B, Cc, 2, dtype=torch.float32)
        y = self.new(B, HW, Cc, dtype=torch.float32)
        a = L.F32Args()
        a.X, a.ws, a.B, a.HW, a.C, a.groups, a.nblk, a.eps, a.silu = x.data_ptr(), part.data_ptr(), B, HW, Cc, groups, nblk, float(eps), int(silu)
        self._f32(L.F32_GN_STATS, a, descr + ".stats")
        a.gamma, a.beta, a.Y = gamma.data_ptr(), beta.data_ptr(), tab.data_ptr()
        self._f32(L.F32_GN_TABLE, a, descr + ".table")
        a.ws, a.Y = tab.data_ptr(), y.data_ptr()
        self._f32(L.F32_GN_APPLY, a, descr + ".apply")
        self.free(part); self.free(tab)
        return y

    def f32_softmax(self, a_, out, scale, descr="f32.softmax"):
        """out[r, :] = softmax(scale * a_[r, :]), fp32 rows"""
        self._chk(a_, descr + ".a", torch.float32); self._chk(out, descr + ".out", torch.float32)
        a = L.F32Args()
        a.X, a.Y, a.M, a.N, a.ldx, a.ldy, a.scale = a_.data_ptr(), out.data_ptr(), a_.shape[0], a_.shape[1], a_.stride(0), out.stride(0), float(scale)
        self._f32(L.F32_SOFTMAX, a, descr)
        return out

    def attention_small(self, q, k, v, B, H, Lq, Lk, dq, dv, scale, out=None, descr="attention_small"):
        """q [B*Lq, H*dq], k [B*Lk, H*dq], v [B*Lk, H*dv] row-major (any row stride) -> [B*Lq, H*dv]"""
        if out is None:
            out = self.new(B * Lq, H * dv)
        a = L.SmallAttnArgs()
        a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
        a.B, a.H, a.Lq, a.Lk, a.dq, a.dv = B, H, Lq, Lk, dq, dv
        a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        a.scale, a.dtype = scale, self.dt
        self._emit(L.OP_ATTN_SMALL, a, descr=descr, flops=2.0 * B * H * Lq * Lk * (dq + dv), keep=(q, k, v, out))
        return out

    # ------------------------------------------------------------------ norms
    def _norm_args(self, B, HW, Cc, groups, eps, silu, mode):
        a = L.NormArgs()
        a.B, a.HW, a.C, a.groups, a.eps, a.silu, a.dtype, a.mode = B, HW, Cc, groups, eps, int(silu), self.dt, mode
        return a

    def gn_stats(self, x, sub=10, descr="gn_stats"):
        """stand-alone GroupNorm statistics of x [B, HW, C] (a pass over x) in the hand-over format -- for tensors whose writing
        launch has no statistics epilogue (conv_in, split-K / ring variants)"""
        self._chk(x, descr + ".x")
        B, HW, Cc = x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
        if Cc % sub:
            raise L.ImhError(f"{descr}: sub-run width {sub} does not divide C={Cc}")
        nblk = self.lib.imh_groupnorm_stats_blocks(HW, Cc)
        t = self.new(B, nblk, Cc // sub, 2, dtype=torch.float32)
        a = self._norm_args(B, HW, Cc, 1, 1.0, 0, L.GN_STATS)
        a.x, a.partial, a.sub = x.data_ptr(), t.data_ptr(), sub
        self._emit(L.OP_GROUPNORM, a, descr=descr, nbytes=float(x.numel() * x.element_size()), keep=(x, t))
        return GnStats(t, nblk, sub, 0, Cc)

    def gn_table(self, stats, gamma, beta, groups, eps, HW, descr="gn_table"):
        """(scale, shift) table [B, C, 2] fp32 of a GroupNorm from the partials of its input: stats = GnStats or a list of two
        (channel concat [a | b]: C = a.C + b.C)"""
        srcs = list(stats) if isinstance(stats, (list, tuple)) else [stats]
        if not 1 <= len(srcs) <= 2:
            raise L.ImhError(f"{descr}: one or two statistics sources")
        Cc = sum(g.C for g in srcs)
        B = srcs[0].t.shape[0]
        cpg = Cc // groups
        for g in srcs:
            if tuple(g.t.shape) != (B, g.nblk, g.C // g.sub, 2) or g.t.dtype != torch.float32 or cpg % g.sub or srcs[0].C % g.sub:
                raise L.ImhError(f"{descr}: statistics {tuple(g.t.shape)} (sub {g.sub}) do not fit C={Cc}, groups={groups}")
        tab = self.new(B, Cc, 2, dtype=torch.float32)
        a = self._norm_args(B, HW, Cc, groups, eps, 0, L.GN_TABLE)
        a.gamma, a.beta, a.table = self._p(gamma), self._p(beta), tab.data_ptr()
        a.partial, a.nblk, a.sub, a.npart, a.C1 = srcs[0].t.data_ptr(), srcs[0].nblk, srcs[0].sub, srcs[0].npart, srcs[0].C
        if len(srcs) == 2:
            a.partial2, a.nblk2, a.sub2, a.npart2 = srcs[1].t.data_ptr(), srcs[1].nblk, srcs[1].sub, srcs[1].npart
        self._emit(L.OP_GROUPNORM, a, descr=descr, keep=(gamma, beta, tab) + tuple(g.t for g in srcs))
        return tab

    def gn_apply(self, x, tab, silu, out=None, descr="gn_apply"):
        """y = silu?(x * scale + shift) as a pass (the consumers that cannot take the table themselves: Linear layers, the
        implicit-GEMM convs)"""
        self._chk(x, descr + ".x")
        B, HW, Cc = x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
        if tuple(tab.shape) != (B, Cc, 2) or tab.dtype != torch.float32:
            raise L.ImhError(f"{descr}: table {tuple(tab.shape)} does not fit [{B}, {Cc}, 2]")
        if out is None:
            out = self.new(*x.shape)
        a = self._norm_args(B, HW, Cc, 1, 1.0, silu, L.GN_APPLY)
        a.x, a.y, a.table = x.data_ptr(), out.data_ptr(), tab.data_ptr()
        es = x.element_size()
        self._emit(L.OP_GROUPNORM, a, descr=descr, flops=4.0 * x.numel(), nbytes=2.0 * es * x.numel(), keep=(x, out, tab))
        return out

    def gn_table_apply(self, x, spec, silu, out=None, descr="gn_table_apply"):
        """y = silu?(GroupNorm(x)) in ONE launch from the producers' partials (IMH_GN_TABLE_APPLY): every workgroup of the apply pass
        builds its sample's table in LDS -- the form for consumers that cannot take the norm themselves (Transformer2DModel.norm in
        front of proj_in, conv_norm_out)"""
        self._chk(x, descr + ".x")
        B, HW, Cc = x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
        spec.check(B, Cc, descr)
        if out is None:
            out = self.new(*x.shape)
        a = self._norm_args(B, HW, Cc, spec.groups, spec.eps, silu, L.GN_TABLE_APPLY)
        a.x, a.y, a.gamma, a.beta = x.data_ptr(), out.data_ptr(), self._p(spec.gamma), self._p(spec.beta)
        s0 = spec.srcs[0]
        a.partial, a.nblk, a.sub, a.npart, a.C1 = s0.t.data_ptr(), s0.nblk, s0.sub, s0.npart, s0.C
        if len(spec.srcs) == 2:
            s1 = spec.srcs[1]
            a.partial2, a.nblk2, a.sub2, a.npart2 = s1.t.data_ptr(), s1.nblk, s1.sub, s1.npart
        es = x.element_size()
        self._emit(L.OP_GROUPNORM, a, descr=descr, flops=4.0 * x.numel(), nbytes=2.0 * es * x.numel(), keep=(x, out) + spec.tensors())
        return out

    def groupnorm(self, x, gamma, beta, groups, eps, silu, out=None, descr="groupnorm", stats=None):
        """x: [B, HW, C] (NHWC flattened).  stats = GnStats left by the producing launch's epilogue (gemm(gn_out=...) /
        conv3x3(gn_groups=...)) or by gn_stats(): table + apply, no statistics pass over x; None: statistics + table + apply."""
        self._chk(x, descr + ".x")
        B, HW, Cc = x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
        if stats is not None:
            if not isinstance(stats, GnStats) or stats.C != Cc or stats.t.shape[0] != B:
                raise L.ImhError(f"{descr}: statistics do not fit x {tuple(x.shape)}")
            return self.gn_table_apply(x, GnSpec(stats, gamma, beta, groups, eps), silu, out=out, descr=descr)
        if out is None:
            out = self.new(*x.shape)
        a = self._norm_args(B, HW, Cc, groups, eps, silu, L.GN_ALL)
        a.x, a.y, a.gamma, a.beta = x.data_ptr(), out.data_ptr(), self._p(gamma), self._p(beta)
        # scratch use is confined to this op's three kernels (stream order), so the shared workspace is safe
        a.partial = self.workspace(self.lib.imh_groupnorm_workspace_bytes(B, HW, Cc, groups)).data_ptr()
        es = x.element_size()
        self._emit(L.OP_GROUPNORM, a, descr=descr, flops=8.0 * x.numel(), nbytes=3.0 * es * x.numel(), keep=(x, out, gamma, beta))
        return out

    def layernorm(self, x, gamma, beta, eps, out=None, descr="layernorm"):
        self._chk(x, descr + ".x")
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        if out is None:
            out = self.new(*x.shape)
        a = L.NormArgs()
        a.x, a.y, a.gamma, a.beta = x.data_ptr(), out.data_ptr(), self._p(gamma), self._p(beta)
        a.rows, a.C, a.eps, a.dtype = rows, Cc, eps, self.dt
        es = x.element_size()
        self._emit(L.OP_LAYERNORM, a, descr=descr, flops=8.0 * x.numel(), nbytes=2.0 * es * x.numel(),
                   keep=(x, out, gamma, beta))
        return out

    # ------------------------------------------------------------------ elementwise
    def ew(self, op, y, a=None, b=None, w=None, bias=None, tab=None, step=None, n=0, i=(0, 0, 0, 0, 0, 0),
           f=(0.0, 0.0, 0.0, 0.0), descr="ew", nbytes=0.0):
        e = L.EwArgs()
        e.a, e.b, e.y, e.w, e.bias = self._p(a), self._p(b), y.data_ptr(), self._p(w), self._p(bias)
        e.tab, e.step = self._p(tab), self._p(step)
        e.n = n
        e.i0, e.i1, e.i2, e.i3, e.i4, e.i5 = i
        e.f0, e.f1, e.f2, e.f3 = f
        e.dtype = self.dt
        self._emit(L.OP_EW, e, ew_op=op, descr=descr, nbytes=nbytes, keep=(a, b, y, w, bias, tab, step))
        return y

    def silu(self, x, descr="silu"):
        out = self.new(*x.shape)
        return self.ew(L.EW_SILU, out, a=x, n=x.numel(), descr=descr, nbytes=2.0 * x.numel() * x.element_size())

    def concat(self, a, b, descr="concat"):
        """NHWC channel concat: a [..., C1], b [..., C2] -> [..., C1 + C2]."""
        c1, c2 = a.shape[-1], b.shape[-1]
        pix = a.numel() // c1
        out = self.new(*a.shape[:-1], c1 + c2)
        return self.ew(L.EW_CONCAT, out, a=a, b=b, n=pix, i=(c1, c2, 0, 0, 0, 0), descr=descr,
                       nbytes=2.0 * out.numel() * out.element_size())

    # ------------------------------------------------------------------ weight prefetch
    def _cold(self, tensors):
        """operands that are parameters (not pool-owned activations) and big enough to matter: every layer's
        weights are read once per forward, i.e. always from HBM unless a previous kernel prefetches them"""
        out = []
        for t in tensors[:2]:
            if t is None or t.numel() * t.element_size() < (256 << 10):
                continue
            if t.untyped_storage().data_ptr() in self._bases:
                continue
            out.append((t.data_ptr(), min(t.numel() * t.element_size(), 0xFFFFFFF0)))
        return out

    PF_CHUNK, PF_CAP_ONLY, PF_BACK = 0, False, 3
    # Round 6: a weight matrix larger than PF_BIG is NOT handed to the launch before it (only its first PF_BIG_CAP bytes, 0 = none).  The one
    # such matrix of the UNet is ff.net.0's [10240, 1280] (26 MB): prefetching it at the tail of the cross-attention's to_out cost that launch
    # 3.8 us (19.4 vs 15.7 us -- the "4 us slower behind the fused cross-attention than behind the self-attention" of round 5's trace) and
    # bought ff.net.0 nothing: its K loops hide their own weight stream (62.2 vs 62.1 us).  -0.28 ms per forward, 559.5 / 561.7 -> 551.6 /
    # 551.3 ms per 30-step denoise in alternating processes (profiles/r06_forward_ab_prefetch_big.json, r06_bench_ab_prefetch_big.txt).
    # Conv weights of that size keep their prefetch: the LDS-halo kernels' two-slot weight rings do not hide a cold stream (8192 x 640 x 17280:
    # 202.9 -> 233.8 us without).  IMH_PF_BIG=0 restores the round-5 behaviour (A/B).
    PF_BIG, PF_BIG_CAP = int(os.environ.get("IMH_PF_BIG", str(14 << 20))), int(os.environ.get("IMH_PF_BIG_CAP", "0"))

    def finalize_prefetch(self):
        """Give every recorded launch the weights of the launch that follows it (tail_prefetch in the kernels):
        a cold weight matrix goes to the nearest earlier op (up to 3 back) whose prefetch slot is still free."""
        if self._pf_done or not self.record:
            return
        self._pf_done = True
        can = (L.OP_GEMM, L.OP_ATTN, L.OP_LAYERNORM, L.OP_GROUPNORM, L.OP_GEMM_DUAL, L.OP_XATTN)

        def carries(i):
            kind, a = self._ops[i][0], self._ops[i][1]
            if kind not in can:
                return False
            # of the GroupNorm launches only the apply pass touches its prefetch slot (gn_apply_kernel); a table / statistics launch
            # would swallow the pointer and the cold conv weights behind it would never be prefetched
            if kind == L.OP_GROUPNORM and a.mode not in (L.GN_ALL, L.GN_APPLY, L.GN_TABLE_APPLY):
                return False
            return True

        taken = set()
        # PF_CHUNK (bytes; A/B, tools/forward_ab.py `pf_chunk`): a weight matrix larger than this is handed out in pieces -- the first to the
        # nearest earlier launch with a free slot, the next to the one before it, ... (up to PF_BACK launches back); PF_CAP: pieces beyond
        # the first are dropped instead (what is not prefetched is read cold)
        chunk, cap_only, back = self.PF_CHUNK, self.PF_CAP_ONLY, self.PF_BACK
        for j, (kind, args, cold) in enumerate(self._ops):
            for (ptr, nb) in cold:
                pieces = [(ptr, nb)]
                is_conv = kind == L.OP_GEMM and bool(getattr(args, "conv", 0))
                if self.PF_BIG and nb > self.PF_BIG and not is_conv:      # a Linear weight this large: only its first PF_BIG_CAP bytes (0 = none) are prefetched
                    pieces = [(ptr, self.PF_BIG_CAP)] if self.PF_BIG_CAP > 0 else []
                elif chunk and nb > chunk:
                    pieces = [(ptr + o, min(chunk, nb - o)) for o in range(0, nb, chunk)]
                    if cap_only:
                        pieces = pieces[:1]
                # nearest earlier launch with a free slot (measured better than handing big matrices to a longer,
                # earlier kernel: 27.0 vs 27.3 ms per forward)
                i = j - 1
                for (pp, pn) in pieces:
                    while i > max(j - 1 - back, -1) and (i in taken or not carries(i)):
                        i -= 1
                    if i <= max(j - 1 - back, -1):
                        break
                    a = self._ops[i][1]
                    tgt = a[0] if self._ops[i][0] == L.OP_GEMM_DUAL else a
                    tgt.pf_ptr, tgt.pf_bytes = pp, pn
                    ref = C.cast(a, C.c_void_p) if self._ops[i][0] == L.OP_GEMM_DUAL else C.byref(a)
                    L.check(self.lib.imh_plan_update(self.plan, i, ref), "imh_plan_update")
                    taken.add(i)
                    i -= 1
        return len(taken)

    # ------------------------------------------------------------------ plans
    def run(self):
        self.finalize_prefetch()
        L.check(self.lib.imh_plan_run(self.plan, self.stream()), "imh_plan_run")

    def capture(self):
        """Capture the recorded plan into a hipGraph (needs a non-default stream)."""
        self.finalize_prefetch()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            L.check(self.lib.imh_plan_capture(self.plan, C.c_void_p(s.cuda_stream)), "imh_plan_capture")
        torch.cuda.current_stream(self.device).wait_stream(s)
        self.captured = True

    def replay(self):
        if self.captured:
            L.check(self.lib.imh_plan_replay(self.plan, self.stream()), "imh_plan_replay")
        else:
            self.run()

    def time_ops(self):
        self.finalize_prefetch()
        n = self.lib.imh_plan_size(self.plan)
        ms = (C.c_float * n)()
        L.check(self.lib.imh_plan_time_ops(self.plan, self.stream(), ms, n), "imh_plan_time_ops")
        return list(ms)

    def __del__(self):
        try:
            if self.plan:
                self.lib.imh_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass
