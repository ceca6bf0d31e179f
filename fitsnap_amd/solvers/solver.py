"""Solver base class: the reference's plugin contract (fitsnap3lib/solvers/solver.py:19-48,
137, 368-435) on top of the MI355X statistics engine.

Contract kept from the reference
  * ``__init__(name, pt, config, linear=True)``; attributes ``fit, errors, linear, cov,
    fit_sam, configs, df, a, b, w, residuals, weighted, all_fits``;
  * ``perform_fit(a=None, b=None, w=None, fs_dict=None, trainall=False)`` returns None and
    sets ``self.fit`` (ndarray (K,), float64) ON RANK 0 ONLY; mask precedence
    ``fs_dict`` > ``trainall`` > ``pt.fitsnap_dict['Testing']``; with no arrays the data
    comes from ``pt.shared_arrays['a'|'b'|'w'].array``;
  * explicit-array quirk of the reference (svd.py:46, ridge.py:39): ``w`` is multiplied
    into ``a[training]`` WITHOUT being masked, so ``w`` must already have one entry per
    training row (or be broadcastable); anything else raises numpy's broadcast ValueError;
  * ``error_analysis(a=None, b=None, w=None, fs_dict=None)``, ``fit_gather()``.

What is different underneath: A, b stay resident in HBM; mask x weight x A^T A runs as
one fused fp64-MFMA kernel per GPU; with several ranks (one per GPU) every rank fits its
OWN rows and the K x K statistics are summed with one RCCL all-reduce before rank 0
solves.  There is no numpy fallback: without the HIP library / a gfx950 device
``perform_fit`` raises.
"""
from __future__ import annotations

import numpy as np

from .. import _capi
from .._hostblas import blas_threads


class _TrainWeights:
    """Weights of the TRAINING rows only, as the reference's explicit-array call hands them over (svd.py:46), plus what
    the device needs to spread them over all rows: the uint8 mask and its exclusive prefix sum.  ``cached`` = the mask
    arrays come from the solver's cache (a re-weighting loop), so the device copy of an earlier call may still be valid."""

    __slots__ = ("w", "mask_u8", "rank", "idx", "cached")

    def __init__(self, w, mask_u8, rank, idx, cached):
        self.w, self.mask_u8, self.rank, self.idx, self.cached = w, mask_u8, rank, idx, cached

    def full(self):
        out = np.zeros(self.mask_u8.shape[0])
        out[self.idx] = self.w
        return out


def _digest(buf) -> bytes:
    """128-bit digest of a bytes-like object: xxh3 when the module is there (10-20 GB/s), else blake2b (~1 GB/s)."""
    try:
        import xxhash

        return xxhash.xxh3_128_digest(buf)
    except ImportError:                      # pragma: no cover - xxhash ships with this image
        import hashlib

        return hashlib.blake2b(buf, digest_size=16).digest()


def _is_label_container(x) -> bool:
    return isinstance(x, (list, np.ndarray)) or (hasattr(x, "codes") and hasattr(x, "categories"))


def _content_stamp(labels):
    """Fingerprint of the whole content of one row-label container (see ``Solver._labels_stamp``)."""
    version = getattr(labels, "version", None)
    if version is not None and isinstance(labels, list):
        # parallel_tools.LabelList counts its own edits: (object, edit count, length) pins the content in O(1).  The
        # caches keep the list object alive (``_cache_hit`` compares by identity first), so the address cannot be recycled
        return ("counted", id(labels), version, len(labels))
    if isinstance(labels, np.ndarray):
        if labels.dtype != object:
            arr = labels if labels.flags.c_contiguous else np.ascontiguousarray(labels)
            return (labels.shape, labels.dtype.str, _digest(memoryview(arr.reshape(-1).view(np.uint8))))
        labels = labels.tolist()
    codes = getattr(labels, "codes", None)
    if codes is not None and hasattr(labels, "categories"):            # pandas.Categorical
        return ("categorical", len(labels), tuple(labels.categories), _content_stamp(np.asarray(codes)))
    n = len(labels)
    if n and isinstance(labels[0], str):
        try:
            joined = "\0".join(labels)
            # a label that itself holds the separator could collide with a different split of the same text
            # (['a\0', 'b'] vs ['a', '\0b']): such lists take the generic walk
            if joined.count("\0") == n - 1:
                return (n, "str", _digest(joined.encode()))
        except TypeError:                    # mixed entries: the generic walk
            pass
    return (n, hash(tuple(labels)))


_EPS = float(np.finfo(np.float64).eps)


RCOND_MARGIN = 10.0
"""``rcond`` below = ``fsnap_solve``'s ``rcond_est`` for the LSTSQ kinds: min(smallest pivot of the Jacobi-scaled Cholesky,
lambda_min of the scaled matrix S estimated FROM THE FACTOR by a few Lanczos steps on S^-1 -- csrc/fsnap_condest.h).  Both
are estimates from ABOVE; the Lanczos one is within ~1.3 x of lambda_min wherever the statistics resolve it, the pivot alone
can be off by a factor exponential in K (A = Z (I - triu(1, 1)), K = 26: pivot 0.04, lambda_min 1e-16 -- round 5 decided
from the pivot and returned answers 1e-6 ... 1 from lstsq on that family).  Every rule divides by this margin first."""


def refinement_skip(K, rcond):
    """No refinement step at all: the normal-equation solve is already far inside the parity bar.

    S has a unit diagonal, so lambda_max <= K and kappa(S) <= K / lambda_min; the solve from the statistics is accurate to
    ~kappa(S) eps.  Below 1e-10 -- four decades inside the 1e-6 bar of the reference's own checker
    (tests/example_checker.py:54-62) -- a pass over the rows buys nothing a caller can see."""
    return rcond is not None and rcond > 0.0 and K * _EPS * RCOND_MARGIN / rcond < 1.0e-10


def refinement_done(K, step, prev_step, beta_max, rcond):
    """After a refinement step of size ``step`` = max|delta| (the one before: ``prev_step``; before the first step: max|beta|):
    stop when what is LEFT is below the accuracy the reference's lstsq itself has, ~kappa(A_w) eps = eps / sqrt(rcond).

    Refinement with the row residual contracts the error by rho ~ kappa^2 eps per step; rho is taken as the larger of
    the prediction K eps / lambda_min and the contraction just observed (step / prev_step), so what is left after this step is
    ~ rho / (1 - rho) * step.  On the benchmark problem (kappa ~ 1e4: kappa^2 eps ~ 2e-8) the first correction is ~1e-8 |beta|
    and the second would be ~1e-16 |beta|: one pass over the rows instead of two.  A slowly converging system (rho >= 1/2)
    never stops early; the absolute floor 1e-14 |beta| is the old rule."""
    if not step > 1.0e-14 * beta_max:
        return True
    if rcond is None or not rcond > 0.0 or not prev_step > 0.0:
        return False
    lam = rcond / RCOND_MARGIN
    rho = max(K * _EPS / lam, step / prev_step)
    if rho >= 0.5:
        return False
    return rho / (1.0 - rho) * step <= max(4.0 * _EPS / np.sqrt(rcond), 1.0e-14) * beta_max


class Solver:
    """Base class for linear solvers (see module docstring)."""

    MAX_DEVICE_CATEGORIES = 3000     # (group, train/test, row type) categories fsnap_error_stats_k keeps in LDS

    def __init__(self, name, pt, config, linear=True):
        self.config = config
        self.pt = pt
        self.name = name
        self.configs = None
        self.fit = None
        self.all_fits = None
        self.template_error = False
        self.errors = []
        self.weighted = "Unweighted"
        self.residuals = None
        self.a = None
        self.b = None
        self.w = None
        self._df = None              # error_analysis DataFrame, built on first access (see the df property)
        self._df_parts = None
        self._cat_cache = None       # (label lists, content stamp, m, category ids, group keys) of the last error_analysis
        self._cat_ctx = None         # context that holds those category ids on the device
        self._err_layout = None      # (group keys, index, source rows, weighting flags) of the last errors table
        self._all_idx = None         # (group keys, [(sub key, member indices)]) of the last *ALL merge
        self._mask_cache = None      # (Testing list, content stamp, training mask, derived arrays) (keep_resident only)
        self._all_train = None       # (bool ones, the same bytes as uint8) of the last `trainall` fit
        self.trust_label_version = False   # keep_resident: key the label caches on pt.labels_version instead of the content
        self.linear = linear
        self.cov = None
        self.fit_sam = None
        # engine state
        self.keep_resident = False   # explicit-array mode: reuse the HBM copy of (a, b) across calls
        self.refine_steps = 0        # iterative-refinement steps after the K x K solve (SVD sets 2)
        self._resident_key = None
        self._stats_host = None      # (G, c, scalars) of the last fit, host ndarrays (after all-reduce)
        self._stats_dev = None       # or (ctx, device address, K) while they are still in HBM only
        self.device_error_stats = True  # error_analysis: grouped reductions on the GPU (False = pandas groupby)
        self.last_rank = None
        self.last_rcond = None
        self.last_refine_steps = 0   # refinement steps the last fit actually took (<= refine_steps)
        self.last_row_space = None   # diagnostics of the last row-space solve (passes, deviation, ...), or None
        self._checks()

    # ------------------------------------------------------------------------------
    def perform_fit(self):
        """Base class function for performing a fit."""
        pass

    def fit_gather(self):
        pass

    def prepare_data(self, a, b, w, fs_dict):
        """``aw, bw`` = the training rows of ``a`` and ``b`` multiplied by their weights (reference:
        fitsnap3lib/solvers/solver.py:50-76), by the stand-alone wavefront weighting kernel (``fsnap_weight_rows``:
        HBM-bound, one 4-row group per wave) on the resident rows; the rows of the testing set are dropped on the host.
        ``fs_dict = None`` trains on all rows; explicit arrays take one weight per TRAINING row like the reference
        (``w[:, np.newaxis] * a[training]``, solver.py:75), ``a = b = w = None`` takes the shared arrays (the reference's
        own last line indexes ``a`` unconditionally and raises there; its evident intent is what happens here).
        Products of two doubles: bit-identical to numpy."""
        a_, b_, w_full, mask, shared_mode = self._resolve_inputs(a, b, w, fs_dict, fs_dict is None)
        if np.ndim(a_) != 2:
            raise ValueError("the A matrix must be 2-D")
        ctx = self._upload(a_, b_, shared_mode)
        self._push_weights(ctx, w_full, mask)
        aw, bw = ctx.weight_rows()
        if not mask.all():
            keep = mask.astype(bool)
            aw, bw = aw[keep], bw[keep]
        return aw, bw

    def _checks(self):
        # fitsnap3lib/solvers/solver.py:106-107
        calc = self.config.sections["CALCULATOR"]
        assert not (calc.linear and calc.per_atom_energy and self.config.args.perform_fit), \
            "Can only output per_atom_energy for non-linear fits (e.g., Pytorch) or with the '--nofit' flag. " \
            "Either change to a non-linear fit, use the flag '--nofit', or set per_atom_energy = 0."

    # ------------------------------------------------------------------------------
    # mask / weights exactly as the reference resolves them
    # ------------------------------------------------------------------------------
    def _labels_stamp(self, lists):
        """What a cache derived from row-label containers is valid for (``keep_resident`` only): a fingerprint of their
        whole CONTENT, so that an in-place edit of ANY entry (one configuration moved between folds) is seen, as in the
        reference, which re-reads the labels on every call.  (A sampled probe, rounds 1-2, missed edits of less than
        ~0.4 % of a list.)  What the walk costs depends on the container the caller hands over (10^6 rows):

        * ``numpy.ndarray`` (bool / integer / fixed-width string): xxh3 over the array's buffer, no copy -- 0.1 ms for a
          bool array, 1.7 ms for 12 MB of ``<U3`` group names;
        * ``pandas.Categorical``: its integer codes + the category tuple -- 0.1 ms whatever the names look like;
        * Python ``list``: every entry is visited -- ``hash(tuple(lst))`` (9 ms for bools), strings through one joined
          buffer (12 ms instead of 26 ms).

        A caller that wants even that gone promises not to edit the containers without saying so:
        ``solver.trust_label_version = True`` keys the caches on the identity of the objects plus ``pt.labels_version``,
        which ``pt.touch_labels()`` bumps."""
        if self.trust_label_version:
            return ("version", getattr(self.pt, "labels_version", 0)) + tuple(id(l) for l in lists)
        return ("content",) + tuple(_content_stamp(l) for l in lists)

    def _cache_hit(self, cached_lists, cached_stamp, lists):
        """Same list objects (the cache keeps them alive, so an address cannot be recycled) with the same content."""
        return (cached_lists is not None and len(cached_lists) == len(lists)
                and all(c is l for c, l in zip(cached_lists, lists))
                and cached_stamp == self._labels_stamp(lists))

    def invalidate_row_caches(self):
        """Forget everything derived from the row-label lists (training mask, category ids, table layouts)."""
        self._mask_cache = None
        self._cat_cache = None
        self._cat_ctx = None
        self._err_layout = None
        self._all_idx = None

    def _training_mask(self, a, fs_dict, trainall):
        """svd.py:35-40 / ridge.py:28-33."""
        if fs_dict is not None:
            lst = fs_dict["Testing"]
            if self.keep_resident and _is_label_container(lst):
                # re-weighting loops pass the same (large) label list every time: what is derived from it (row indices,
                # prefix sums, the mask resident on the GPU) is kept while the list's content stays the same.  The cache
                # holds the list itself and a stamp of its whole content; without keep_resident nothing is cached.
                mc = self._mask_cache
                if mc is not None and self._cache_hit((mc[0],), mc[1], (lst,)):
                    return mc[2]
                mask = ~np.asarray(lst, dtype=bool)
                self._mask_cache = (lst, self._labels_stamp((lst,)), mask, None)
                return mask
            return ~np.asarray(lst, dtype=bool)
        if trainall:
            return self._all_train_mask(np.shape(a)[0])[0]
        # after Calculator.collect_distributed_lists the dictionary holds the lists of ALL ranks; the
        # rows of this rank's arrays are described by the local copy kept in pt.local_lists
        local = getattr(self.pt, "local_lists", None)
        if local and "Testing" in local:
            return ~np.asarray(local["Testing"], dtype=bool)
        return ~np.asarray(self.pt.fitsnap_dict["Testing"], dtype=bool)

    def _all_train_mask(self, m):
        """(bool mask, uint8 mask) of ``m`` rows that all train -- built once per row count: a `trainall` fit of 10^6 rows
        otherwise spends ~0.2 ms in numpy (ones, astype, count_nonzero, all) before it reaches the GPU."""
        at = self._all_train
        if at is None or at[0].shape[0] != m:
            ones = np.ones(m, dtype=bool)
            ones.flags.writeable = False                   # shared between calls: nobody may turn a row off in place
            at = self._all_train = (ones, ones.view(np.uint8))
        return at

    def _is_all_train(self, mask):
        at = self._all_train
        return at is not None and (mask is at[0] or mask is at[1])

    def _resolve_inputs(self, a, b, w, fs_dict, trainall):
        """Returns (a, b, w_full, mask_u8, shared_mode)."""
        pt = self.pt
        training = self._training_mask(a, fs_dict, trainall)
        if a is None and b is None and w is None:
            sa, sb, sw = pt.shared_arrays["a"], pt.shared_arrays["b"], pt.shared_arrays["w"]
            a, b, w_full = sa.array, sb.array, sw.array
            if len(training) != a.shape[0]:
                raise IndexError("boolean index did not match indexed array along axis 0; size of axis is "
                                 f"{a.shape[0]} but size of corresponding boolean axis is {len(training)}")
            mask_u8 = self._all_train[1] if self._is_all_train(training) else training.astype(np.uint8)
            return a, b, np.asarray(w_full, dtype=np.float64), mask_u8, True
        a = np.asarray(a)
        b = np.asarray(b)
        w = np.asarray(w, dtype=np.float64)
        m = a.shape[0]
        if len(training) != m:
            raise IndexError("boolean index did not match indexed array along axis 0; size of axis is "
                             f"{m} but size of corresponding boolean axis is {len(training)}")
        # a cached mask (re-weighting loop) carries its uint8 form and count along -- and, once a fit asked for them, the
        # row indices and prefix sum: aux = [idx | None, mask_u8, rank | None, ntrain].  The SAME mask_u8 object comes back
        # on every call, so that identity-keyed residency checks downstream (``_push_weights``) can hit
        mc = self._mask_cache
        cached = mc is not None and mc[2] is training
        aux = mc[3] if cached else None
        if aux is None and self._is_all_train(training):
            aux = [None, self._all_train[1], None, m]
        if aux is None:
            mask_u8 = training.astype(np.uint8)
            aux = [None, mask_u8, None, int(np.count_nonzero(mask_u8))]
            if cached:
                self._mask_cache = (mc[0], mc[1], training, aux)
        idx, mask_u8, rank, ntrain = aux

        def train_aux():
            # row indices and exclusive prefix sum of the mask: only a fit that hands over one weight per TRAINING row of a
            # partly masked matrix needs them (5 ms of numpy at 10^6 rows: not on the path of `trainall` / full-weight calls)
            nonlocal idx, rank
            if idx is None:
                rk = np.cumsum(mask_u8, dtype=np.int64) - mask_u8
                idx, rank = np.flatnonzero(training), rk.astype(np.int32)
                aux[0], aux[2] = idx, rank                  # in place: the cache entry (if any) holds this very list

        # reference: aw = w[:, None] * a[training]  (numpy broadcasting on the row axis)
        if w.ndim == 0 or w.shape[0] == 1:
            w_full = np.full(m, float(w.reshape(-1)[0]))
        elif w.shape[0] == ntrain:
            if ntrain == m:
                w_full = w
            else:
                # one weight per training row: the GPU spreads them over the rows (fsnap_set_weights_train)
                had = cached and idx is not None
                train_aux()
                w_full = _TrainWeights(w, mask_u8, rank, idx, had)
        else:
            raise ValueError(f"operands could not be broadcast together with shapes ({w.shape[0]},1) ({ntrain},{a.shape[1]}) ")
        return a, b, w_full, mask_u8, False

    # ------------------------------------------------------------------------------
    # the hot path
    # ------------------------------------------------------------------------------
    def _push_weights(self, ctx, w_full, mask):
        """Row weights and training mask of this fit onto the device."""
        if isinstance(w_full, _TrainWeights):
            if w_full.cached and getattr(ctx, "resident_train_mask", None) is w_full.mask_u8:
                try:
                    ctx.set_weights_train(w_full.w)             # mask and prefix sum of an earlier call are resident
                    return
                except _capi.FsnapError:
                    pass                                        # somebody replaced them meanwhile: send them again
            ctx.set_weights_train(w_full.w, w_full.mask_u8, w_full.rank)
            return
        ctx.set_weights(w_full, None if (self._is_all_train(mask) or mask.all()) else mask)

    def _upload(self, a, b, shared_mode):
        ctx = self.pt.hip()
        if shared_mode:
            sa, sb = self.pt.shared_arrays["a"], self.pt.shared_arrays["b"]
            key = ("shared", id(sa), getattr(sa, "version", 0), id(sb), getattr(sb, "version", 0), a.shape)
            # rows assembled on the device (Calculator.flush_rows) are already resident
            if (getattr(sa, "device_version", -1) == getattr(sa, "version", 0)
                    and getattr(sb, "device_version", -1) == getattr(sb, "version", 0)
                    and ctx.m == a.shape[0] and ctx.K == a.shape[1]):
                self._resident_key = key
                return ctx
            # shared arrays are re-uploaded unless their owner kept the version (touch())
            if self.keep_resident and key == self._resident_key:
                return ctx
        else:
            key = ("explicit", a.__array_interface__["data"][0], a.shape, a.strides,
                   b.__array_interface__["data"][0]) if isinstance(a, np.ndarray) and isinstance(b, np.ndarray) else None
            if self.keep_resident and key is not None and key == self._resident_key:
                return ctx
        ctx.upload_rows(a, b)
        self._resident_key = key
        return ctx

    def _local_statistics(self, a, b, w_full, mask, shared_mode):
        """This rank's (G, c, scalars) on the host."""
        ctx = self._upload(a, b, shared_mode)
        self._push_weights(ctx, w_full, mask)
        return ctx.normal_eq()

    def _fit_statistics(self, a=None, b=None, w=None, fs_dict=None, trainall=False):
        """mask x weight x normal equations on this rank's GPU, summed over ranks.

        Returns (G, c, scalars) as host ndarrays on every rank (scalars = [bw.bw, sum(bw),
        n_train]); also stored in ``self.last_statistics``."""
        pt = self.pt
        a, b, w_full, mask, shared_mode = self._resolve_inputs(a, b, w, fs_dict, trainall)
        if a.ndim != 2:
            raise ValueError("the A matrix must be 2-D")
        K = a.shape[1]
        if a.shape[0] > 0:
            G, c, s = self._local_statistics(a, b, w_full, mask, shared_mode)
        else:
            G, c, s = np.zeros((K, K)), np.zeros(K), np.zeros(3)
        if pt.multi:
            host = np.concatenate([np.asarray(G, dtype=np.float64).ravel(), c, s])
            pt.allreduce_statistics(host)
            G = host[:K * K].reshape(K, K).copy()
            c = host[K * K:K * K + K].copy()
            s = host[K * K + K:].copy()
        self.last_statistics = (G, c, s)
        return G, c, s

    LAPACK_FALLBACK_K = 256     # above this the truncating fallback of a K x K solve runs in LAPACK (numpy.linalg.eigh)

    def _solve(self, kind, param, G, c):
        """Host K x K solve on statistics already on the host."""
        K = len(c)
        probe = _capi.PROBE_OF.get(kind)
        if probe is not None and K > self.LAPACK_FALLBACK_K:
            beta, rank, rcond = _capi.solve(probe, param, G, c)
            if rank < 0 and kind == _capi.SOLVE_RIDGE_INV:
                # regressor.py:15 itself: np.linalg.inv raises LinAlgError on a singular matrix, as the reference does
                beta, rank = np.linalg.inv(np.asarray(G) + float(param) * np.eye(K)) @ np.asarray(c), K
            elif rank < 0:
                beta, rank = self._truncated_eigen_solve(kind, param, G, c)
        else:
            beta, rank, rcond = _capi.solve(kind, param, G, c)
        self.last_rank, self.last_rcond = rank, rcond
        return beta

    @staticmethod
    def _truncated_eigen_solve(kind, param, G, c):
        """The library's fallback for statistics no Cholesky resolves (``fsnap_solve``: eigenvalues below the cut
        dropped -- gelsd's truncation seen through G for LSTSQ, sklearn Ridge's SVD fallback for RIDGE), with LAPACK's
        symmetric eigensolver in place of the library's cyclic Jacobi sweeps, which take O(K^3) each on one core
        (58 s at K = 1595 against ~1 s here).  Returns (beta, rank)."""
        G = np.asarray(G, dtype=np.float64)
        c = np.asarray(c, dtype=np.float64)
        K = len(c)
        ridge = kind in (_capi.SOLVE_RIDGE, _capi.SOLVE_RIDGE_PROBE)
        alpha = float(param) if ridge else 0.0
        # exactly-zero columns (zero row and column of a Gram matrix, c_j = 0): beta_j = 0 with or without a ridge term;
        # left in, LAPACK's tridiagonalisation smears their unit eigenvectors over the other near-null directions
        keep = np.diag(G) != 0.0
        beta = np.zeros(K)
        n = int(np.count_nonzero(keep))
        if n == 0:
            return beta, 0
        M = 0.5 * (G[np.ix_(keep, keep)] + G[np.ix_(keep, keep)].T)
        M[np.diag_indices(n)] += alpha
        with blas_threads(n):
            ev, V = np.linalg.eigh(M)
        eps = np.finfo(np.float64).eps
        cut = 4.0 * n * eps if ridge else max(float(param) ** 2 if param > 0 else 0.0, 4.0 * n * eps)
        used = ev > cut * np.max(np.abs(ev))
        Vu = V[:, used]
        beta[keep] = Vu @ ((Vu.T @ c[keep]) / ev[used])
        return beta, int(np.count_nonzero(used))

    def _fit_and_solve(self, kind, param, a=None, b=None, w=None, fs_dict=None, trainall=False):
        """The latency path of SVD / RIDGE: the statistics stay in HBM, the K x K system is solved through
        ``fsnap_solve_device`` (page-locked mirror + host factorisation for small K, blocked Cholesky on the GPU for
        large K) and only beta crosses PCIe; ``last_statistics`` is fetched lazily.

        Multi-GPU job on the native RCCL transport: ``fsnap_fit_dist`` -- kernel, in-place all-reduce on the same
        stream, solve -- on EVERY rank (deterministic solve of bit-identical sums: the ranks agree on beta, rank and
        conditioning without a broadcast; ``perform_fit`` still publishes ``fit`` on rank 0 only, as the reference).
        A ``torch.distributed`` group (CPU tests) reduces the statistics through the host instead."""
        pt = self.pt
        if pt.multi and pt.comm_kind != "rccl":
            G, c, _ = self._fit_statistics(a, b, w, fs_dict, trainall)
            return self._solve(kind, param, G, c)
        a, b, w_full, mask, shared_mode = self._resolve_inputs(a, b, w, fs_dict, trainall)
        if a.ndim != 2:
            raise ValueError("the A matrix must be 2-D")
        K = a.shape[1]
        have_rows = a.shape[0] > 0
        self._stats_host = None
        self._stats_dev = None
        if not have_rows and not pt.multi:
            self._stats_host = (np.zeros((K, K)), np.zeros(K), np.zeros(3))
            return self._solve(kind, param, *self._stats_host[:2])
        ctx = pt.hip()
        if have_rows:
            ctx = self._upload(a, b, shared_mode)
            self._push_weights(ctx, w_full, mask)
        elif ctx.m > 0:
            ctx.drop_rows()                 # rows of an EARLIER fit must not stand in for this rank's empty share
            self._resident_key = None
        if pt.multi:
            beta, rank, rcond, ptr = ctx.fit_dist(kind, param, K)       # collective
        else:
            beta, rank, rcond, ptr = ctx.fit_resident(kind, param)      # kernel + reduction + K x K solve, one library call
        self._stats_dev = (ctx, ptr, K)
        self.last_rank, self.last_rcond = rank, rcond
        return beta

    def _resolve_probe(self, kind, param, beta):
        """After a ``*_PROBE`` solve that came back unresolved (rank -1) and with no better path to take: finish with the
        truncating solve on the downloaded statistics (LAPACK for large K).  Same on every rank."""
        if self.last_rank is None or self.last_rank >= 0:
            return beta
        G, c, _ = self.last_statistics
        base = _capi.BASE_OF[kind]
        rcond = self.last_rcond
        beta = self._solve(base, param, G, c)
        self.last_rcond = rcond if rcond is not None else self.last_rcond
        return beta

    def _rows_on_device(self):
        """True when this rank's context holds rows (a rank of a multi-GPU job may own none)."""
        ctx = self.pt._hip
        return ctx is not None and ctx.m > 0

    def _refine(self, beta, kind, param, steps):
        """Iterative refinement of a least-squares / ridge solution with the residual formed
        from the ROWS (``fsnap_residual_rhs``: s = (wA)^T (wb - wA beta), one streaming pass
        over the resident A): G delta = s - alpha beta, beta += delta.  Takes the error of the
        normal-equation solve from ~kappa^2 eps to ~kappa eps (measured on the golden Ta set:
        7e-8 -> 5e-13 vs the reference lstsq).  At most ``steps`` steps; how many are TAKEN follows the conditioning
        the Cholesky reported (``refinement_skip`` / ``refinement_done``) and is kept in ``last_refine_steps``.
        Collective in a multi-rank job: the right-hand side is all-reduced and every rank solves the same system with
        the same ``last_rcond``, so all ranks take the same number of steps."""
        pt = self.pt
        alpha = param if kind in (_capi.SOLVE_RIDGE, _capi.SOLVE_RIDGE_INV) else 0.0
        G = None
        self.last_refine_steps = 0
        rcond = self.last_rcond
        if refinement_skip(len(beta), rcond):
            return beta
        prev = float(np.max(np.abs(beta))) if len(beta) else 0.0
        for _ in range(int(steps)):
            s = pt._hip.residual_rhs(beta)[0] if self._rows_on_device() else np.zeros(len(beta))
            if pt.multi:
                pt.allreduce_host(s)
            rhs = s - alpha * beta
            if self._stats_dev is not None and self._stats_host is None:
                # statistics still in HBM: solve there (large K: blocked Cholesky on the GPU, G never crosses PCIe)
                dctx, dptr, dK = self._stats_dev
                delta, rank, _ = dctx.solve_device(kind, param, dK, dptr, rhs=rhs)
            else:
                if G is None:
                    G = self.last_statistics[0]
                delta, rank, _ = _capi.solve(kind, param, G, rhs)
            if rank < len(beta):        # truncated (rank-deficient) solve: refinement is not meaningful
                break
            beta = beta + delta
            self.last_refine_steps += 1
            step = float(np.max(np.abs(delta)))
            if refinement_done(len(beta), step, prev, float(np.max(np.abs(beta))), rcond):
                break
            prev = step
        return beta

    ROWSPACE_RCOND = 1.0e-11    # lambda_min of the Jacobi-scaled statistics (estimate / RCOND_MARGIN) below which they are not trusted

    def _needs_row_space(self, K):
        """After a LSTSQ solve from the statistics: did the K x K system resolve the problem?  No when lambda_min of the
        scaled matrix -- ``last_rcond`` / ``RCOND_MARGIN``, see there -- is below ``ROWSPACE_RCOND`` (kappa of the
        equilibrated A_w beyond ~3e5: the refinement with the normal matrix stops converging around 1e7) or when the
        solve dropped directions that are not simply exactly-zero columns.  Same answer on every rank (the inputs are
        the all-reduced statistics)."""
        rank, rcond = self.last_rank, getattr(self, "last_rcond", None)
        if rank is None or rcond is None:
            return False
        if rank < 0:            # a probe that came back unresolved: no need to look at the statistics
            return True
        ill = rcond / RCOND_MARGIN < self.ROWSPACE_RCOND
        if rank < K:
            G = self.last_statistics[0]
            zero_cols = int(np.count_nonzero(np.diag(G) == 0.0))
            return rank < K - zero_cols or ill
        return ill

    def _row_space_fit(self, K, rcond):
        """``lstsq(aw, bw, rcond)`` on the rows (``fsnap_lstsq_rows``): CholeskyQR passes on the GPU, dgelsd's K x K end
        on the host.  The rows and weights of the fit that just ran are resident.  Collective in a multi-rank job."""
        ctx = self.pt.hip()
        if not self.pt.multi and self._stats_dev is not None and self._stats_dev[0] is ctx and self._stats_dev[2] == K:
            # the fit from the statistics that sent us here ran on these rows, weights and mask a moment ago: the first
            # CholeskyQR pass starts from its Gram matrix instead of computing it again (one-shot, see include/fsnap_hip.h)
            ctx.set_option("rowspace_reuse_stats", 1)
        beta, rank, info = ctx.lstsq_rows(rcond, K)
        self.last_rank = rank
        self.last_row_space = info
        return beta

    @property
    def last_statistics(self):
        """(G, c, scalars) of the last fit as host ndarrays (downloaded on first access when the
        fit kept them in HBM)."""
        if self._stats_host is None and self._stats_dev is not None:
            ctx, ptr, K = self._stats_dev
            self._stats_host = ctx.download_packed(ptr, K)
        return self._stats_host

    @last_statistics.setter
    def last_statistics(self, value):
        self._stats_host = value
        self._stats_dev = None

    # ------------------------------------------------------------------------------
    # downstream of the fit (solver.py:108-133, 368-435)
    # ------------------------------------------------------------------------------
    def _offset(self):
        """SNAP with ``bzeroflag``: the fit has no constant term, the potential file wants one per type -- a zero B0
        goes in front of every type's block of ``ncoeff`` coefficients (solver.py:78-103).  Shapes as in the
        reference: several types -> column vector, one type -> flat vector; ``fit_sam`` (coefficient samples of the
        UQ solvers, one row per sample) gets the same zeros."""
        bis = self.config.sections["BISPECTRUM"]
        ntypes, ncoeff = bis.numtypes, bis.ncoeff

        def with_b0(rows):
            blocks = np.asarray(rows, dtype=np.float64).reshape(-1, ntypes, ncoeff)
            return np.pad(blocks, ((0, 0), (0, 0), (1, 0))).reshape(blocks.shape[0], ntypes * (ncoeff + 1))

        if ntypes > 1:
            self.fit = with_b0(self.fit).reshape(-1, 1)
            if self.fit_sam is not None:
                self.fit_sam = with_b0(self.fit_sam)
        else:
            self.fit = np.insert(self.fit, 0, 0)
            if self.fit_sam is not None:
                self.fit_sam = np.insert(self.fit_sam, 0, 0, axis=1)

    @staticmethod
    def _host_error_sums(truths, preds, weights, cat, ncat):
        """The ten sums of ``fsnap_error_stats`` (include/fsnap_hip.h) for rows that are on the host only (CPU process
        groups, ``device_error_stats = False``): one ``bincount`` per sum.  Row ``c`` of the result belongs to
        category ``c``; ``_metrics_from_sums`` turns it into the numbers of solver.py:108-133."""
        t = np.asarray(truths, dtype=np.float64)
        w = np.asarray(weights, dtype=np.float64)
        r = t - np.asarray(preds, dtype=np.float64)
        cat = np.asarray(cat, dtype=np.int64)

        def per_cat(x):
            return np.bincount(cat, weights=x, minlength=ncat)

        n = np.bincount(cat, minlength=ncat).astype(np.float64)
        nw = per_cat((w != 0).astype(np.float64))
        st, swt = per_cat(t), per_cat(w * t)
        with np.errstate(divide="ignore", invalid="ignore"):
            mean_t = np.where(n > 0, st / n, 0.0)
            mean_wt = np.where(nw > 0, swt / nw, 0.0)
        return np.stack([n, nw, st, swt, per_cat(np.abs(r)), per_cat(r * r), per_cat((t - mean_t[cat]) ** 2),
                         per_cat(np.abs(w * r)), per_cat((w * r) ** 2), per_cat((w * t - mean_wt[cat]) ** 2)], axis=1)

    def _host_error_tables(self, df):
        """(per-group table, *ALL table) from a DataFrame with truths / preds / weights and the three label columns."""
        gb = df.groupby(["Groups", "Testing", "Row_Type"], sort=True, observed=True)
        keys = list(gb.size().index)
        st = self._host_error_sums(df["truths"].to_numpy(), df["preds"].to_numpy(), df["weights"].to_numpy(),
                                   gb.ngroup().to_numpy(), len(keys))
        return self._tables_from_sums(keys, st)

    @staticmethod
    def _metrics_from_sums(n, nw, st, swt, sar, srr, sct, sawr, swrr, scwt):
        """The eight numbers of solver.py:119-133 from the ten sums of fsnap_error_stats (arrays, one entry per
        group).  Same quirks: w_mae divides by ALL rows, w_rmse by the rows with non-zero weight."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return {"ncount": n, "mae": sar / n, "rmse": np.sqrt(srr / n), "rsq": 1 - srr / sct,
                    "w_ncount": nw, "w_mae": sawr / n, "w_rmse": np.sqrt(swrr / nw), "w_rsq": 1 - swrr / scwt}

    @staticmethod
    def _pool_sums(rows):
        """Pools the ten sums of fsnap_error_stats of several disjoint row sets (rows: (k, 10) array) into those of
        their union.  Counts and plain sums add; the centred sums are re-centred on the pooled mean:
        sum (x - M)^2 = sum_c [ S_c + 2 (mu_c - M) (sum x_c - n_c mu_c) + n_c (mu_c - M)^2 ]."""
        rows = np.asarray(rows, dtype=np.float64).reshape(-1, 10)
        n, nw, s_t, s_wt = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
        N, NW = n.sum(), nw.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            mu_c, M = s_t / n, s_t.sum() / N
            wmu_c = np.where(nw > 0, s_wt / np.where(nw > 0, nw, 1), 0.0)
            WM = s_wt.sum() / NW
        sct = np.sum(rows[:, 6] + 2 * (mu_c - M) * (s_t - n * mu_c) + n * (mu_c - M) ** 2)
        scwt = np.sum(rows[:, 9] + 2 * (wmu_c - WM) * (s_wt - n * wmu_c) + n * (wmu_c - WM) ** 2)
        return np.array([N, NW, s_t.sum(), s_wt.sum(), rows[:, 4].sum(), rows[:, 5].sum(), sct, rows[:, 7].sum(),
                         rows[:, 8].sum(), scwt])

    _METRIC_COLUMNS = ("ncount", "mae", "rmse", "rsq", "w_ncount", "w_mae", "w_rmse", "w_rsq")

    def _metric_arrays(self, keys, st):
        """((len(keys), 8) per-group metrics, (n_all, 8) *ALL metrics) of solver.py:391-405 from the (len(keys), 10) array
        of sums, columns as in ``_METRIC_COLUMNS``; what depends on the key list only (the two row indexes, the members
        of every *ALL row) is built once per list and kept in ``_all_idx``."""
        from pandas import MultiIndex

        st = np.asarray(st, dtype=np.float64).reshape(len(keys), 10)
        if self._all_idx is None or self._all_idx[0] is not keys:
            subs = sorted({(k[1], k[2]) for k in keys})
            self._all_idx = (keys, [(tk, np.array([i for i, k in enumerate(keys) if (k[1], k[2]) == tk])) for tk in subs],
                             MultiIndex.from_tuples(keys, names=["Groups", "Testing", "Row_Type"]),
                             MultiIndex.from_tuples(subs, names=["Testing", "Row_Type"]))
        gm = self._metrics_from_sums(*(st[:, k] for k in range(10)))
        # *ALL rows: pool the groups of one (Testing, Row_Type)
        pooled = np.array([self._pool_sums(st[idx]) for _, idx in self._all_idx[1]]).reshape(len(self._all_idx[1]), 10)
        am = self._metrics_from_sums(*(pooled[:, k] for k in range(10)))
        cols = self._METRIC_COLUMNS
        return (np.column_stack([np.asarray(gm[c], dtype=np.float64) for c in cols]),
                np.column_stack([np.asarray(am[c], dtype=np.float64) for c in cols]))

    def _tables_from_sums(self, keys, st):
        """(per-group table, *ALL table) of solver.py:391-405 from the (len(keys), 10) array of sums."""
        from pandas import DataFrame

        gm, am = self._metric_arrays(keys, st)
        cols = list(self._METRIC_COLUMNS)
        return DataFrame(gm, index=self._all_idx[2], columns=cols), DataFrame(am, index=self._all_idx[3], columns=cols)

    def _device_error_sums(self, a, b, w, shared, fs_dict):
        """(sorted group keys, (len(keys), 10) sums) of the rows of THIS rank from fsnap_error_stats."""
        ctx = self._upload(a, np.asarray(b), shared)        # no copy when these rows are already resident
        ctx.set_weights(np.asarray(w, dtype=np.float64))
        cat, keys, fresh = self._row_categories(fs_dict, np.shape(a)[0])   # sorted group keys, position = category id
        beta = np.asarray(self.fit, dtype=np.float64).reshape(-1)
        # the category ids stay on the device between calls; (context, serial) tells whether they are still OURS
        mine = (not fresh) and self._cat_ctx is not None and self._cat_ctx == (id(ctx), getattr(ctx, "cat_serial", -1))
        if len(keys) > self.MAX_DEVICE_CATEGORIES:
            # more categories than the kernel's LDS table holds: predictions from the GPU GEMV, grouping on the host
            preds, _ = ctx.predict(beta)
            return keys, self._host_error_sums(b, preds, w, cat, len(keys))
        try:
            st = ctx.error_stats(beta, None if mine else cat, len(keys))
        except _capi.FsnapError:
            st = ctx.error_stats(beta, cat, len(keys))       # rows were replaced meanwhile: send the ids again
        self._cat_ctx = (id(ctx), getattr(ctx, "cat_serial", -1))
        return keys, st

    def _device_error_tables(self, a, b, w, shared, fs_dict):
        """(per-group table, *ALL table) of solver.py:391-405, built from fsnap_error_stats."""
        keys, st = self._device_error_sums(a, b, w, shared, fs_dict)
        return self._tables_from_sums(keys, st)

    def _global_keys(self, keys):
        """Sorted union over the ranks of the (group, testing, row type) keys.  The exchange is one small all-gather of
        Python objects; a re-weighting loop (``keep_resident``) that keeps presenting the same local key list reuses
        the union -- its per-candidate traffic is then the fixed-size table of doubles alone.  Collective: every rank
        must take the same branch, which holds when all ranks run the same loop with ``keep_resident`` set alike."""
        gk = getattr(self, "_gkeys_cache", None)
        if self.keep_resident and gk is not None and gk[0] is keys:
            return gk[1]
        union = sorted({k for part in self.pt.allgather_object(list(keys)) for k in part})
        self._gkeys_cache = (keys, union)
        return union

    def _allgather_tables(self, table):
        """(ranks, len(table), 10) array of every rank's table of sums (equal shapes: rows follow the global keys)."""
        pt = self.pt
        table = np.ascontiguousarray(table, dtype=np.float64)
        if pt.comm_kind == "rccl":
            blobs = pt.hip().allgather_bytes(table.tobytes() or b"\0", pt._size)
            return np.array([np.frombuffer(q[:table.nbytes], dtype=np.float64).reshape(table.shape) for q in blobs])
        return np.array(pt.allgather_object(table))

    def _merge_rank_sums(self, parts):
        """Multi-GPU error analysis: every rank reduced ITS rows to (keys, sums); the union over the ranks is the
        table of all rows (a group may live on several ranks: its sums are pooled).  parts: [(keys, st), ...]."""
        allkeys = sorted({k for keys, _ in parts for k in keys})
        pos = {k: i for i, k in enumerate(allkeys)}
        buckets = [[] for _ in allkeys]
        for keys, st in parts:
            st = np.asarray(st, dtype=np.float64).reshape(len(keys), 10)
            for k, row in zip(keys, st):
                buckets[pos[k]].append(row)
        merged = np.array([self._pool_sums(np.array(rows)) for rows in buckets]).reshape(len(allkeys), 10)
        return allkeys, merged

    @property
    def df(self):
        """The reference's per-row DataFrame (descriptor columns, truths, preds, weights, row labels; solver.py:376-389).
        Built on first access: a fit loop that only reads ``errors`` (the GA of examples/library/genetic_algorithm) never
        pays for it."""
        if self._df is None and self._df_parts is not None:
            from pandas import DataFrame

            a, b, w, fs_dict, shared = self._df_parts
            df = DataFrame(a)
            df["truths"] = np.asarray(b).tolist()
            if self.fit is not None:
                fit = np.asarray(self.fit, dtype=np.float64).reshape(-1)
                if fit.shape[0] == np.shape(a)[1]:
                    df["preds"] = self.predict_rows() if shared else self.predict_rows(a, b)
            df["weights"] = np.asarray(w).tolist()
            for key in fs_dict.keys():
                if _is_label_container(fs_dict[key]) and len(fs_dict[key]) == len(df.index):
                    df[key] = list(fs_dict[key]) if isinstance(fs_dict[key], list) else fs_dict[key]
            self._df = df
        return self._df

    @df.setter
    def df(self, value):
        self._df = value
        self._df_parts = None

    def _row_categories(self, fs_dict, m):
        """Category id of every row = index of its (Groups, Testing, Row_Type) key in sorted order (the order of the
        reference's groupby).  Like the reference the labels are re-read on every call; only with ``keep_resident``
        (re-weighting loops that pass the same label lists every time) are the ids cached -- keyed on the list
        objects themselves plus a stamp of their whole content (``_labels_stamp``)."""
        from pandas import DataFrame

        lists = (fs_dict["Groups"], fs_dict["Testing"], fs_dict["Row_Type"])
        cc = self._cat_cache
        if (self.keep_resident and cc is not None and cc[2] == m and all(_is_label_container(l) for l in lists)
                and self._cache_hit(cc[0], cc[1], lists)):
            return cc[3], cc[4], False
        gb = DataFrame({"Groups": lists[0], "Testing": lists[1], "Row_Type": lists[2]}).groupby(
            ["Groups", "Testing", "Row_Type"], sort=True, observed=True)     # observed: Categorical labels list only what occurs
        cat = gb.ngroup().to_numpy(dtype=np.int32)
        keys = list(gb.size().index)
        stamp = self._labels_stamp(lists) if self.keep_resident else None
        self._cat_cache = (lists, stamp, m, cat, keys)
        return cat, keys, True

    def _assemble_errors(self, grouped, allrows, layout_key=None):
        """The reference's errors table (solver.py:391-429) from the per-group and *ALL metric tables (8 columns each:
        unweighted and weighted ncount / mae / rmse / rsq).  The pandas reshaping (two concats, level reordering,
        sorting, relabelling) costs ~3 ms and depends only on the group keys: with ``layout_key`` (the key list of a
        re-weighting loop, compared by identity) the row order is derived once -- by pushing row tags through the very
        same pandas pipeline -- and later tables are filled by one numpy gather."""
        from pandas import DataFrame, concat

        names = ["ncount", "mae", "rmse", "rsq"]
        cols = [names, ["w_ncount", "w_mae", "w_rmse", "w_rsq"]]
        ren = dict(zip(cols[1], names))

        def pipeline(g, a):
            g = concat({"Unweighted": g[cols[0]], "weighted": g[cols[1]].rename(columns=ren)},
                       names=["Weighting"]).reorder_levels(["Groups", "Weighting", "Testing", "Row_Type"]).sort_index()
            a = concat({"Unweighted": a[cols[0]], "weighted": a[cols[1]].rename(columns=ren)},
                       names=["Weighting"]).reorder_levels(["Weighting", "Testing", "Row_Type"]).sort_index()
            e = concat([concat({"*ALL": a}, names=["Groups"]), g])
            e.index.rename(["Group", "Weighting", "Testing", "Subsystem"], inplace=True)
            e.index = e.index.set_levels(["Testing" if t else "Training" for t in e.index.levels[2]], level=2)
            return e

        if layout_key is not None:
            lay = self._err_layout
            if isinstance(grouped, np.ndarray):
                # metric arrays of a re-weighting loop whose layout is known: one gather, one DataFrame
                vals = np.vstack([allrows, grouped])[lay[2]]
                errors = DataFrame(np.where(lay[3][:, None], vals[:, 4:], vals[:, :4]), index=lay[1], columns=names)
                errors.ncount = errors.ncount.astype(int)
                return errors
            if lay is None or lay[0] is not layout_key or lay[4] != (len(allrows), len(grouped)):
                # tags: source row (the *ALL rows first) in the unweighted columns, -(source row) - 1 in the weighted ones
                na, ng = len(allrows), len(grouped)
                ta = DataFrame({c: (np.arange(na) if c in names else -np.arange(na) - 1) for c in cols[0] + cols[1]},
                               index=allrows.index, dtype=np.int64)
                tg = DataFrame({c: (np.arange(ng) + na if c in names else -(np.arange(ng) + na) - 1) for c in cols[0] + cols[1]},
                               index=grouped.index, dtype=np.int64)
                tagged = pipeline(tg, ta)
                tag = tagged["ncount"].to_numpy()
                weighted = tag < 0
                src = np.where(weighted, -tag - 1, tag)
                lay = self._err_layout = (layout_key, tagged.index, src, weighted, (na, ng))
            vals = np.vstack([allrows[cols[0] + cols[1]].to_numpy(dtype=np.float64),
                              grouped[cols[0] + cols[1]].to_numpy(dtype=np.float64)])[lay[2]]
            out = np.where(lay[3][:, None], vals[:, 4:], vals[:, :4])
            errors = DataFrame(out, index=lay[1], columns=names)
        else:
            errors = pipeline(grouped, allrows)
        errors.ncount = errors.ncount.astype(int)
        return errors

    def predict_rows(self, a=None, b=None):
        """``preds = a @ self.fit`` (solver.py:377) on the GPU (streaming GEMV kernel)."""
        if a is None:
            a, b = self.pt.shared_arrays["a"].array, self.pt.shared_arrays["b"].array
            ctx = self._upload(a, b, True)
        else:
            a = np.asarray(a)
            if b is None:
                b = np.zeros(a.shape[0])
            ctx = self._upload(a, np.asarray(b), False)
        preds, _ = ctx.predict(np.asarray(self.fit, dtype=np.float64).reshape(-1))
        return preds

    def error_analysis(self, a=None, b=None, w=None, fs_dict=None):
        """Linear part of the reference's error analysis (solver.py:368-435): per
        (group, train/test, row type) weighted and unweighted count / MAE / RMSE / R^2,
        then the bzeroflag offset.  Predictions come from the GPU GEMV kernel."""
        from pandas import DataFrame

        self.errors = []
        pt = self.pt
        multi = pt.multi
        shared = a is None and b is None and w is None and fs_dict is None
        if multi:
            # One process per GPU: every rank owns the rows of ITS configurations.  The fit lives on
            # rank 0 (reference semantics); broadcast it, evaluate the local rows on every GPU and bring
            # only what the table needs to rank 0.  The reference sees all rows on rank 0 through the
            # node-shared array instead.
            self.fit = pt.bcast_object(self.fit, src=0)
            if shared:
                a = pt.shared_arrays["a"].array
                b = pt.shared_arrays["b"].array
                w = pt.shared_arrays["w"].array
                local = pt.local_lists if getattr(pt, "local_lists", None) else pt.fitsnap_dict
            else:
                local = fs_dict
            # Rows resident on every rank's GPU: each rank reduces ITS rows to the (groups x 10) table of sums
            # (fsnap_error_stats); the ranks agree on the union of their group keys, then ONE fixed-size all-gather of
            # doubles brings the tables to rank 0, which pools them -- no per-row gather ("reduce
            # per-group error sums").  The per-row DataFrame is not built in this mode (EXTRAS dump_dataframe takes
            # the gather below).
            if (self.device_error_stats and self.fit is not None and not self.config.sections["EXTRAS"].dump_dataframe
                    and not self.config.sections["SOLVER"].true_multinode and pt._transport.on_gpu):
                keys, st = self._device_error_sums(a, b, w, shared, local) if len(b) > 0 else ([], np.zeros((0, 10)))
                gkeys = self._global_keys(keys)
                table = np.zeros((len(gkeys), 10))
                if len(keys):
                    pos = {k: i for i, k in enumerate(gkeys)}
                    table[[pos[k] for k in keys]] = np.asarray(st, dtype=np.float64).reshape(len(keys), 10)
                tables = self._allgather_tables(table)
                if pt._rank != 0:
                    self.fit = None
                    return
                self._df = None
                self._df_parts = None
                merged = np.array([self._pool_sums(rows[rows[:, 0] > 0]) for rows in np.swapaxes(tables, 0, 1)])
                grouped, allrows = self._tables_from_sums(gkeys, merged.reshape(len(gkeys), 10))
                self.errors = self._assemble_errors(grouped, allrows, None)
                if (self.config.sections["CALCULATOR"].calculator == "LAMMPSSNAP"
                        and "BISPECTRUM" in self.config.sections and self.config.sections["BISPECTRUM"].bzeroflag):
                    self._offset()
                return
            preds = (self.predict_rows() if shared else self.predict_rows(a, b)) if self.fit is not None else None
            n = len(b)
            piece = {"truths": np.asarray(b), "preds": preds, "weights": np.asarray(w),
                     "lists": {k: list(v) for k, v in local.items() if _is_label_container(v) and len(v) == n}}
            parts = pt.allgather_object(piece)
            if pt._rank != 0:
                self.fit = None
                return
            self.df = DataFrame({"truths": np.concatenate([q["truths"] for q in parts]).tolist()})
            if self.fit is not None:
                self.df["preds"] = np.concatenate([q["preds"] for q in parts])
            self.df["weights"] = np.concatenate([q["weights"] for q in parts]).tolist()
            for key in parts[0]["lists"]:
                if all(key in q["lists"] for q in parts):
                    self.df[key] = [x for q in parts for x in q["lists"][key]]
        else:
            if pt._rank != 0:
                return
            if shared:
                a = pt.shared_arrays["a"].array
                b = pt.shared_arrays["b"].array
                w = pt.shared_arrays["w"].array
                fs_dict = pt.fitsnap_dict
            self._df = None
            self._df_parts = (a, b, w, fs_dict, shared)      # the DataFrame itself is built on first access
        if self.config.sections["EXTRAS"].dump_dataframe:
            self.df.to_pickle(self.config.sections["EXTRAS"].dataframe_file)
        if self.fit is not None and not self.config.sections["SOLVER"].true_multinode:
            if not multi and self.device_error_stats:
                # single GPU: the rows are resident -- predictions and the grouped reductions run on the GPU
                # (fsnap_error_stats); only the (groups x 10) table of sums comes back
                keys, st = self._device_error_sums(a, b, w, shared, fs_dict)
                lay = self._err_layout
                if lay is not None and lay[0] is keys:
                    grouped, allrows = self._metric_arrays(keys, st)        # layout known: no intermediate DataFrames
                    if lay[4] != (len(allrows), len(grouped)):
                        grouped, allrows = self._tables_from_sums(keys, st)
                else:
                    grouped, allrows = self._tables_from_sums(keys, st)
            else:
                grouped, allrows = self._host_error_tables(self.df)
            layout_key = self._cat_cache[4] if (not multi and self.device_error_stats and self._cat_cache is not None) else None
            self.errors = self._assemble_errors(grouped, allrows, layout_key)
        if self.fit is not None:
            if (self.config.sections["CALCULATOR"].calculator == "LAMMPSSNAP"
                    and "BISPECTRUM" in self.config.sections and self.config.sections["BISPECTRUM"].bzeroflag):
                self._offset()
