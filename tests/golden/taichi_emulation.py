"""A small pure-Python/NumPy emulation of the Taichi subset the reference rasteriser uses, so that the reference's
OWN source files can be executed in this container (Taichi is not installed and its kernels are CUDA-only).

TEST INFRASTRUCTURE ONLY, used by tests/golden/make_reference_operator_vectors.py to produce golden vectors; nothing
in the product imports it.  What is emulated, and how faithfully:

* scalars are NumPy float32 / Python ints, so every arithmetic operation rounds to fp32 like Taichi's default
  ``default_fp=f32`` (NumPy 2 keeps float32 when a Python literal is mixed in); no fused multiply-add, no fast-math;
* ``ti.math.vecN / matN``, ``ti.Vector``, ``ti.Matrix``, ``ti.types.vector/matrix``: class ``Mat`` -- element-wise
  ``+ - * /``, ``@`` as an unrolled left-to-right sum of products (Taichi unrolls small matrix products the same way),
  ``transpose / determinant / outer_product / norm / sum / dot``, ``.x .y .z .w``;
* ``@ti.func``: plain call with ``Mat`` arguments copied first (Taichi passes matrices by value: the reference relies on
  it, e.g. utils.py:263-264 adds 0.3 to its argument);  ``@ti.dataclass``: a record class with keyword/positional
  constructor;  ``ti.static``: identity;  ``ti.cast`` to an integer type truncates toward zero;
* ``@ti.kernel``: torch tensors are passed as NumPy views of the same memory; a kernel whose source does not use
  ``ti.simt.block`` runs its loops sequentially; one that does (the two tile blend kernels) is run one 256-thread
  block at a time on real OS threads, ``ti.simt.block.sync()`` being a ``threading.Barrier`` and ``SharedArray`` a
  per-block array, so the barrier-separated staging through shared memory executes as written;
* every TOP-LEVEL loop of a kernel is an offloaded task of its own, as in Taichi, and the tasks run in source order
  (``_OffloadRewriter``): of ``gaussian_point_rasterisation_backward`` the pixel loop runs on the block threads and the
  per-point loop behind it (RAS:707-772) ONCE, after the last tile.  (Until round 4 every pixel thread ran the whole
  body: the same results, that loop only assigns, but pixels x points times its work -- the 2,400-Gaussian vector took
  hours and now takes 90 s.)  ``GS_EMU_PROCS=n`` deals the blocks of a launch to n forked worker processes over
  shared mappings of the array arguments;
* ``ti.atomic_add(a[i], v)`` statements are rewritten (AST) to a locked ``a[i] += v``; the accumulation order is the
  thread (and, with worker processes, block) order, where the GPU's is undefined.
"""
from __future__ import annotations

import ast
import inspect
import os
import sys
import threading
import types

import numpy as np

F32 = np.float32

# ARITHMETIC SWITCHES (round 6, tests/golden/make_arithmetic_residue.py): real Taichi compiles the reference's kernels with
# ``fast_math=True`` -- LLVM may contract a multiplication and an addition into one fused multiply-add and turn a division
# into a multiplication by the reciprocal -- which cannot be observed here.  The default emulation is IEEE fp32 with neither.
# These two switches re-run the reference's sources with one of the liberties taken EVERYWHERE it syntactically can be, to
# measure how much of the reference's output depends on that choice (see load_reference: the rewrite is on the AST).
#   GS_EMU_FMA=1      a*b + c, c + a*b, a*b - c, c - a*b, x += a*b and the sums of products of ``@`` round once
#   GS_EMU_RCP_DIV=1  a / b  is  a * (1 / b), both roundings in fp32
EMU_FMA = os.environ.get("GS_EMU_FMA") == "1"
EMU_RCP_DIV = os.environ.get("GS_EMU_RCP_DIV") == "1"


def _plain_number(v):
    return isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)


def ti_fma(a, b, c):
    """a * b + c with ONE rounding to fp32 for fp32 scalars / vectors (the product of two fp32 numbers is exact in double);
    integers stay integers, anything else (torch tensors of the glue code) takes the plain expression."""
    ops = (a, b, c)
    if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in ops):
        return a * b + c
    if not all(isinstance(v, Mat) or _plain_number(v) for v in ops):
        return a * b + c
    raw = [v.a if isinstance(v, Mat) else v for v in ops]
    r = np.asarray(np.asarray(raw[0], np.float64) * np.asarray(raw[1], np.float64) + np.asarray(raw[2], np.float64)).astype(F32)
    return Mat(r) if any(isinstance(v, Mat) for v in ops) else r[()]


def ti_neg(x):
    return -x


def ti_div(a, b):
    """a / b as a * (1 / b), each step rounded to fp32 (fp32 scalars / vectors only; Taichi's ``/`` on two integers is a
    floating-point division too)."""
    if not all(isinstance(v, Mat) or _plain_number(v) for v in (a, b)):
        return a / b
    ra = a.a if isinstance(a, Mat) else F32(a)
    rb = b.a if isinstance(b, Mat) else F32(b)
    r = ra * (F32(1.0) / rb)
    return Mat(r) if isinstance(r, np.ndarray) else r


# --------------------------------------------------------------------------------------------------- vectors / matrices
def _f(x):
    return x if isinstance(x, (np.floating, np.integer)) and x.dtype == F32 else F32(x)


class Mat:
    __slots__ = ("a",)

    def __init__(self, a):
        self.a = np.array(a, dtype=F32)

    # ---- structure
    @property
    def n(self):
        return self.a.shape[0]

    def __len__(self):
        return self.a.shape[0]

    def __iter__(self):
        if self.a.ndim == 1:
            return iter([self.a[i] for i in range(self.a.shape[0])])
        return iter([Mat(self.a[i]) for i in range(self.a.shape[0])])

    def __getitem__(self, idx):
        r = self.a[idx]
        return Mat(r) if isinstance(r, np.ndarray) else r

    def __setitem__(self, idx, value):
        self.a[idx] = value.a if isinstance(value, Mat) else _f(value)

    def copy(self):
        return Mat(self.a.copy())

    def __repr__(self):
        return f"Mat({self.a!r})"

    # ---- swizzles
    x = property(lambda s: s.a[0], lambda s, v: s.a.__setitem__(0, _f(v)))
    y = property(lambda s: s.a[1], lambda s, v: s.a.__setitem__(1, _f(v)))
    z = property(lambda s: s.a[2], lambda s, v: s.a.__setitem__(2, _f(v)))
    w = property(lambda s: s.a[3], lambda s, v: s.a.__setitem__(3, _f(v)))

    # ---- element-wise arithmetic (fp32)
    @staticmethod
    def _other(o):
        return o.a if isinstance(o, Mat) else _f(o)

    def __add__(self, o): return Mat(self.a + self._other(o))
    def __radd__(self, o): return Mat(self._other(o) + self.a)
    def __sub__(self, o): return Mat(self.a - self._other(o))
    def __rsub__(self, o): return Mat(self._other(o) - self.a)
    def __mul__(self, o): return Mat(self.a * self._other(o))
    def __rmul__(self, o): return Mat(self._other(o) * self.a)
    def __truediv__(self, o): return ti_div(self, o) if EMU_RCP_DIV else Mat(self.a / self._other(o))
    def __rtruediv__(self, o): return ti_div(o, self) if EMU_RCP_DIV else Mat(self._other(o) / self.a)
    def __neg__(self): return Mat(-self.a)
    def __pos__(self): return self

    def __iadd__(self, o):
        self.a = self.a + self._other(o)
        return self

    def __isub__(self, o):
        self.a = self.a - self._other(o)
        return self

    def __imul__(self, o):
        self.a = self.a * self._other(o)
        return self

    # ---- products: unrolled, left to right, every step rounded to fp32
    def __matmul__(self, o):
        if EMU_FMA:
            return self._matmul_fma(o)
        A, B = self.a, o.a
        if A.ndim == 1 and B.ndim == 1:
            s = A[0] * B[0]
            for k in range(1, A.shape[0]):
                s = s + A[k] * B[k]
            return s
        if A.ndim == 1:      # row vector @ matrix
            out = np.empty(B.shape[1], F32)
            for j in range(B.shape[1]):
                s = A[0] * B[0, j]
                for k in range(1, A.shape[0]):
                    s = s + A[k] * B[k, j]
                out[j] = s
            return Mat(out)
        if B.ndim == 1:      # matrix @ column vector
            out = np.empty(A.shape[0], F32)
            for i in range(A.shape[0]):
                s = A[i, 0] * B[0]
                for k in range(1, A.shape[1]):
                    s = s + A[i, k] * B[k]
                out[i] = s
            return Mat(out)
        out = np.empty((A.shape[0], B.shape[1]), F32)
        for i in range(A.shape[0]):
            for j in range(B.shape[1]):
                s = A[i, 0] * B[0, j]
                for k in range(1, A.shape[1]):
                    s = s + A[i, k] * B[k, j]
                out[i, j] = s
        return Mat(out)

    def _matmul_fma(self, o):
        """the same sums of products with every ``s + a*b`` fused (one rounding per step)"""
        A, B = self.a, o.a
        A2 = A.reshape(1, -1) if A.ndim == 1 else A
        B2 = B.reshape(-1, 1) if B.ndim == 1 else B
        out = np.empty((A2.shape[0], B2.shape[1]), F32)
        for i in range(A2.shape[0]):
            for j in range(B2.shape[1]):
                s = A2[i, 0] * B2[0, j]
                for k in range(1, A2.shape[1]):
                    s = ti_fma(A2[i, k], B2[k, j], s)
                out[i, j] = s
        if A.ndim == 1 and B.ndim == 1:
            return out[0, 0]
        if A.ndim == 1:
            return Mat(out[0])
        if B.ndim == 1:
            return Mat(out[:, 0])
        return Mat(out)

    def dot(self, o):
        return self @ o

    def transpose(self):
        return Mat(self.a.T.copy())

    def determinant(self):
        a = self.a
        if a.shape == (2, 2):
            if EMU_FMA:
                return ti_fma(a[0, 0], a[1, 1], -(a[0, 1] * a[1, 0]))
            return a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]
        if a.shape == (3, 3):
            return (a[0, 0] * (a[1, 1] * a[2, 2] - a[2, 1] * a[1, 2]) - a[1, 0] * (a[0, 1] * a[2, 2] - a[2, 1] * a[0, 2]) +
                    a[2, 0] * (a[0, 1] * a[1, 2] - a[1, 1] * a[0, 2]))
        raise NotImplementedError(a.shape)

    def outer_product(self, o):
        out = np.empty((self.a.shape[0], o.a.shape[0]), F32)
        for i in range(self.a.shape[0]):
            for j in range(o.a.shape[0]):
                out[i, j] = self.a[i] * o.a[j]
        return Mat(out)

    def sum(self):
        flat = self.a.reshape(-1)
        s = flat[0]
        for k in range(1, flat.shape[0]):
            s = s + flat[k]
        return s

    def norm(self):
        flat = self.a.reshape(-1)
        s = flat[0] * flat[0]
        for k in range(1, flat.shape[0]):
            s = ti_fma(flat[k], flat[k], s) if EMU_FMA else s + flat[k] * flat[k]
        return np.sqrt(s)


def _flatten(args):
    out = []
    for x in args:
        if isinstance(x, Mat):
            out.extend(_flatten(list(x.a.reshape(-1))))
        elif isinstance(x, (list, tuple)):
            out.extend(_flatten(x))
        elif isinstance(x, np.ndarray):
            out.extend(_flatten(list(x.reshape(-1))))
        else:
            out.append(x)
    return out


class _MatType:
    """vecN / matNxM constructor, also usable as a type annotation."""

    def __init__(self, n, m=None):
        self.shape = (n,) if m is None else (n, m)

    def __call__(self, *args):
        if len(args) == 1 and isinstance(args[0], Mat) and args[0].a.shape == self.shape:
            return args[0].copy()
        flat = _flatten(args)
        size = int(np.prod(self.shape))
        if len(flat) == 1:
            flat = flat * size
        if len(flat) != size:
            raise ValueError(f"cannot build {self.shape} from {len(flat)} values")
        return Mat(np.array([_f(v) for v in flat], F32).reshape(self.shape))


def _vector(values, dt=None):
    return Mat(np.array([_f(v) for v in _flatten([values])], F32))


def _matrix(rows, dt=None):
    return Mat(np.array([[_f(v) for v in _flatten([r])] for r in rows], F32))


def _elementwise(fn):
    def apply(x):
        if isinstance(x, Mat):
            return Mat(fn(x.a))
        return fn(_f(x))
    return apply


def _via_double(fn):
    return lambda a: fn(np.asarray(a, np.float64)).astype(F32)[()]


# exp / log of fp32 arguments.  Default: NumPy's fp32 routines (SIMD polynomials, a few ulps: they differ from the correctly
# rounded value on 39 % of the inputs).  GS_EMU_EXP=cr: evaluated in double and rounded once -- the correctly rounded fp32
# function.  Which exponential the reference's Taichi back end calls is not observable here; the two settings bracket how
# much its results can depend on that choice (nothing visible on ordinary scenes, 1e-4 .. 1e-3 on needle scenes, whose
# conics amplify the last bit of the scale activation a thousandfold: tests/golden/README.md).
_exp, _log = (_via_double(np.exp), _via_double(np.log)) if os.environ.get("GS_EMU_EXP") == "cr" else (np.exp, np.log)


def _is_int(x):
    return isinstance(x, (int, np.integer)) and not isinstance(x, bool)


def _minmax(fn, int_fn):
    def apply(*args):
        r = args[0]
        for a in args[1:]:
            if _is_int(r) and _is_int(a):          # integer operands stay integers (Taichi's type rules)
                r = int_fn(int(r), int(a))
                continue
            ra, aa = (r.a if isinstance(r, Mat) else _f(r)), (a.a if isinstance(a, Mat) else _f(a))
            v = fn(ra, aa)
            r = Mat(v) if isinstance(v, np.ndarray) else v
        return r
    return apply


def _cast(x, t):
    if isinstance(x, Mat):
        return x
    if t in (np.int32, np.int64, np.int8, int):
        return int(x)                      # truncation toward zero, like a C cast
    return F32(x)


def _normalize(v):
    return v / v.norm() if EMU_RCP_DIV else Mat(v.a / v.norm())


# --------------------------------------------------------------------------------------------------- execution model
_tls = threading.local()
_atomic_lock = threading.Lock()


class _NoLock:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_process_lock = _NoLock()     # a multiprocessing lock while a launch is spread over worker processes


class _BlockCtx:
    def __init__(self, dim):
        self.dim = dim
        self.barrier = threading.Barrier(dim)
        self.shared = {}
        self.lock = threading.Lock()
        self.error = None


class _Probe(Exception):
    pass


def _ndrange(*bounds):
    mode = getattr(_tls, "mode", "seq")
    if mode == "probe":
        _tls.probe_n = int(bounds[0])
        raise _Probe()
    if mode == "block":
        return [_tls.tid] if _tls.tid < int(bounds[0]) else []
    if len(bounds) == 1:
        return range(int(bounds[0]))
    import itertools
    return itertools.product(*[range(int(b)) for b in bounds])


def _loop_config(**kw):
    if "block_dim" in kw:
        _tls.block_dim = int(kw["block_dim"])


def _shared_array(shape, dtype=F32):
    ctx = _tls.block
    idx = _tls.shared_calls
    _tls.shared_calls += 1
    with ctx.lock:
        if idx not in ctx.shared:
            ctx.shared[idx] = np.zeros(shape, dtype=np.dtype(dtype))
        return ctx.shared[idx]


def _block_sync():
    _tls.block.barrier.wait()


def ti_atomic_add(array, index, value):
    with _atomic_lock, _process_lock:
        array[index] += value


def _to_numpy(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().numpy()
    except ImportError:  # pragma: no cover
        pass
    return x


def _offload(k):
    """Guards the k-th top-level loop of a kernel (see _OffloadRewriter): True when this execution is to run it."""
    only = getattr(_tls, "offload_only", None)
    return only is None or only == k


def _run_block(fn, call, b, dim, offload=None):
    """One 256-thread block of a tile kernel on real OS threads (barrier = ti.simt.block.sync())."""
    ctx = _BlockCtx(dim)

    def body(k):
        _tls.mode, _tls.tid, _tls.block, _tls.shared_calls, _tls.offload_only = "block", b * dim + k, ctx, 0, offload
        try:
            fn(**call)
        except threading.BrokenBarrierError:
            pass
        except BaseException as exc:  # noqa: BLE001
            ctx.error = exc
            ctx.barrier.abort()
    threads = [threading.Thread(target=body, args=(k,)) for k in range(dim)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if ctx.error is not None:
        raise ctx.error


def _run_blocks_in_processes(fn, call, n_blocks, dim, n_procs, offload=None):
    """The blocks of a tile kernel dealt to ``n_procs`` forked worker processes (GS_EMU_PROCS; a block is still 256 OS
    threads of one process)."""
    def work(w, shared):
        for b in range(w, n_blocks, n_procs):
            _run_block(fn, shared, b, dim, offload)
    _run_in_processes(call, n_procs, work)


def _run_loop_in_processes(fn, call, n_procs, offload):
    """A top-level ``for i in range(n)`` loop of a kernel (a parallel loop in Taichi: its iterations are independent but
    for atomics) dealt to worker processes: worker w runs the iterations w, w + n_procs, ... (``__ti_prange__``)."""
    def work(w, shared):
        _tls.mode, _tls.offload_only, _tls.prange = "seq", offload, (w, n_procs)
        fn(**shared)
    _run_in_processes(call, n_procs, work)


def _prange(r):
    part = getattr(_tls, "prange", None)
    return r if part is None else r[part[0]::part[1]]


def _run_in_processes(call, n_procs, work):
    """``work(w, shared_arguments)`` in ``n_procs`` forked worker processes.  The units of a launch (blocks, loop
    iterations) are independent except for their ``ti.atomic_add``s, whose order the reference leaves undefined: every
    ndarray argument is moved to an anonymous shared mapping for the launch, the atomics take a cross-process lock on
    top of the thread lock, and the arrays are copied back afterwards -- the same statements on the same memory, only
    the interleaving differs (as on a GPU)."""
    import mmap
    import multiprocessing as mp
    import traceback
    global _process_lock
    shared, keep = dict(call), []
    for key, value in call.items():
        if isinstance(value, np.ndarray):
            buf = mmap.mmap(-1, max(value.nbytes, 1))
            view = np.frombuffer(buf, dtype=value.dtype, count=value.size).reshape(value.shape)
            view[...] = value
            shared[key] = view
            keep.append(buf)
    ctx = mp.get_context("fork")
    saved_lock, _process_lock = _process_lock, ctx.Lock()
    sys.stdout.flush()
    sys.stderr.flush()
    pids = []
    try:
        for w in range(n_procs):
            pid = os.fork()
            if pid == 0:
                code = 0
                try:
                    work(w, shared)
                except BaseException:  # noqa: BLE001
                    traceback.print_exc()
                    code = 1
                finally:
                    sys.stdout.flush()
                    sys.stderr.flush()
                    os._exit(code)
            pids.append(pid)
        failed = [pid for pid in pids if os.waitpid(pid, 0)[1] != 0]
        if failed:
            raise RuntimeError(f"{len(failed)} emulation worker(s) failed (traceback above)")
    finally:
        _process_lock = saved_lock
    for key, value in call.items():
        if isinstance(value, np.ndarray):
            value[...] = shared[key]


def _kernel(fn):
    src = inspect.getsource(fn)
    parallel = "simt.block" in src
    sig = inspect.signature(fn)

    def launch(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)
        call = {k: _to_numpy(v) for k, v in bound.arguments.items()}
        info = fn.__globals__.get("__ti_offloads__", {}).get(fn.__name__)
        n_procs = int(os.environ.get("GS_EMU_PROCS", "1"))
        longest = max([v.shape[0] for v in call.values() if isinstance(v, np.ndarray) and v.ndim > 0] + [0])
        if not parallel:
            _tls.mode, _tls.prange = "seq", None
            if info is None or n_procs <= 1 or longest < 4096 or info[2] != info[0]:
                return fn(**call)       # (short loops, or a loop that is not a plain `range`: in this process)
            for k in range(info[0]):    # every top-level loop in its turn, its iterations dealt to the workers
                _run_loop_in_processes(fn, call, n_procs, k)
            _tls.offload_only = None
            return None
        # Every top-level loop of a Taichi kernel is an offloaded task of its own, and the tasks run one after the other.
        # With the kernel's loops numbered by _OffloadRewriter the block loop runs on threads, every other top-level loop
        # ONCE, sequentially, in its place (gaussian_point_rasterisation_backward: the per-point loop RAS:707-772 after
        # the last tile).  Without the numbering (a module not loaded through load_reference) every thread runs the
        # whole body -- the same results as long as the other loops only assign, and pixels x points times the work.
        n_loops, block_loop = info[:2] if info is not None else (1, None)
        for k in range(n_loops):
            only = k if info is not None else None
            if info is not None and k != block_loop:
                if n_procs > 1 and longest >= 4096 and info[2] == n_loops - 1:
                    _run_loop_in_processes(fn, call, n_procs, k)
                else:
                    _tls.mode, _tls.offload_only, _tls.prange = "seq", k, None
                    fn(**call)
                continue
            # how many threads, and the block size: run up to the parallel loop header once
            _tls.mode, _tls.block_dim, _tls.offload_only = "probe", 256, only
            try:
                fn(**call)
                raise RuntimeError("block kernel without an ndrange loop")
            except _Probe:
                pass
            total, dim = _tls.probe_n, _tls.block_dim
            assert total % dim == 0
            if min(n_procs, total // dim) > 1:
                _run_blocks_in_processes(fn, call, total // dim, dim, min(n_procs, total // dim), only)
            else:
                for b in range(total // dim):
                    _run_block(fn, call, b, dim, only)
        _tls.mode, _tls.offload_only = "seq", None
    launch.__wrapped__ = fn
    return launch


def _func(fn):
    def call(*args, **kwargs):
        args = [a.copy() if isinstance(a, Mat) else a for a in args]
        kwargs = {k: (v.copy() if isinstance(v, Mat) else v) for k, v in kwargs.items()}
        return fn(*args, **kwargs)
    call.__wrapped__ = fn
    call.__name__ = getattr(fn, "__name__", "ti_func")
    return call


def _dataclass(cls):
    names = list(getattr(cls, "__annotations__", {}))

    def __init__(self, *args, **kwargs):
        for name, value in zip(names, args):
            setattr(self, name, value)
        for name, value in kwargs.items():
            setattr(self, name, value)
    cls.__init__ = __init__
    return cls


class _Inert:
    """Placeholder for annotation-only API (ti.types.ndarray(...), ti.template(), ...)."""

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self


def build_taichi_module():
    ti = types.ModuleType("taichi")
    tm = types.ModuleType("taichi.math")
    for mod in (tm,):
        mod.vec2, mod.vec3, mod.vec4 = _MatType(2), _MatType(3), _MatType(4)
        mod.mat2, mod.mat3, mod.mat4 = _MatType(2, 2), _MatType(3, 3), _MatType(4, 4)
        mod.exp, mod.sqrt, mod.log = _elementwise(_exp), _elementwise(np.sqrt), _elementwise(_log)
        mod.normalize = _normalize
        mod.dot = lambda a, b: a @ b
        mod.pi = np.pi
    ti.math = tm
    ti.f32 = ti.float32 = np.float32
    ti.i32, ti.i64, ti.i8 = np.int32, np.int64, np.int8
    ti.exp, ti.sqrt, ti.log = tm.exp, tm.sqrt, tm.log
    ti.sin, ti.cos, ti.abs = _elementwise(np.sin), _elementwise(np.cos), _elementwise(np.abs)
    ti.max, ti.min = _minmax(np.maximum, max), _minmax(np.minimum, min)
    ti.cast = _cast
    ti.static = lambda *a: a[0] if len(a) == 1 else a
    ti.Vector, ti.Matrix = _vector, _matrix
    ti.func, ti.kernel, ti.dataclass = _func, _kernel, _dataclass
    ti.ndrange, ti.loop_config = _ndrange, _loop_config
    types_mod = types.ModuleType("taichi.types")
    types_mod.ndarray = _Inert()
    types_mod.vector = lambda n, dtype=None: _MatType(n)
    types_mod.matrix = lambda n, m, dtype=None: _MatType(n, m)
    ti.types = types_mod
    ti.template = _Inert()
    simt, block = types.ModuleType("taichi.simt"), types.ModuleType("taichi.simt.block")
    block.SharedArray, block.sync = _shared_array, _block_sync
    simt.block = block
    ti.simt = simt
    ti.init = lambda *a, **k: None
    rng = np.random.default_rng(0)      # ti.random(): uniform [0, 1) in f32 (the controller's densify-by-sampling, GP3:391-395)
    ti.random = lambda dtype=None: np.float32(rng.random(dtype=np.float32))
    ti.cpu, ti.cuda, ti.gpu = "cpu", "cuda", "gpu"
    ti.profiler = _Inert()
    return ti, tm


# --------------------------------------------------------------------------------------------------- loading the reference
class _AtomicRewriter(ast.NodeTransformer):
    """``ti.atomic_add(X[idx], v)`` (statement) -> ``__ti_atomic_add__(X, idx, v)``."""

    def visit_Expr(self, node):
        self.generic_visit(node)
        c = node.value
        if (isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute) and c.func.attr == "atomic_add" and
                isinstance(c.func.value, ast.Name) and c.func.value.id == "ti" and len(c.args) == 2 and
                isinstance(c.args[0], ast.Subscript)):
            target = c.args[0]
            new = ast.Call(func=ast.Name(id="__ti_atomic_add__", ctx=ast.Load()),
                           args=[target.value, target.slice, c.args[1]], keywords=[])
            return ast.copy_location(ast.Expr(value=new), node)
        return node


class _ArithmeticRewriter(ast.NodeTransformer):
    """GS_EMU_FMA / GS_EMU_RCP_DIV: inside ``@ti.func`` / ``@ti.kernel`` bodies, ``a*b + c`` (either order), ``a*b - c``,
    ``c - a*b`` and ``x += a*b`` become ``__ti_fma__`` calls, ``a / b`` becomes ``__ti_div__(a, b)``.  The torch glue of the
    operator (plain Python between the kernels) is left alone."""

    def __init__(self, fma, rcp_div):
        self.fma, self.rcp_div, self.inside = fma, rcp_div, 0

    @staticmethod
    def _call(name, *args):
        return ast.Call(func=ast.Name(id=name, ctx=ast.Load()), args=list(args), keywords=[])

    def visit_FunctionDef(self, node):
        taichi = any(isinstance(d, ast.Attribute) and d.attr in ("kernel", "func") and isinstance(d.value, ast.Name) and
                     d.value.id == "ti" for d in node.decorator_list)
        self.inside += int(taichi)
        self.generic_visit(node)
        self.inside -= int(taichi)
        return node

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if not self.inside:
            return node
        is_mul = lambda n: isinstance(n, ast.BinOp) and isinstance(n.op, ast.Mult)   # noqa: E731
        new = None
        if self.fma and isinstance(node.op, ast.Add):
            if is_mul(node.left):
                new = self._call("__ti_fma__", node.left.left, node.left.right, node.right)
            elif is_mul(node.right):
                new = self._call("__ti_fma__", node.right.left, node.right.right, node.left)
        elif self.fma and isinstance(node.op, ast.Sub):
            if is_mul(node.left):
                new = self._call("__ti_fma__", node.left.left, node.left.right, self._call("__ti_neg__", node.right))
            elif is_mul(node.right):
                new = self._call("__ti_fma__", self._call("__ti_neg__", node.right.left), node.right.right, node.left)
        elif self.rcp_div and isinstance(node.op, ast.Div):
            new = self._call("__ti_div__", node.left, node.right)
        return ast.copy_location(new, node) if new is not None else node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if (self.inside and self.fma and isinstance(node.op, (ast.Add, ast.Sub)) and isinstance(node.value, ast.BinOp) and
                isinstance(node.value.op, ast.Mult) and isinstance(node.target, ast.Name)):
            load = ast.Name(id=node.target.id, ctx=ast.Load())
            a = node.value.left if isinstance(node.op, ast.Add) else self._call("__ti_neg__", node.value.left)
            return ast.copy_location(ast.Assign(targets=[node.target], value=self._call("__ti_fma__", a, node.value.right, load)),
                                     node)
        return node


class _OffloadRewriter(ast.NodeTransformer):
    """Top-level ``for`` loops of a ``@ti.kernel`` body -> ``if __ti_offload__(k): for ...`` (k = 0, 1, ... in source
    order).  Taichi compiles every top-level loop of a kernel into an offloaded task of its own and runs the tasks in
    order; the launcher (_kernel) uses the numbering to run the loop that uses ``ti.simt.block`` on threads and the other
    loops once each.  ``offloads``: {kernel name: (number of top-level loops, index of the block loop or None, number of plain-range loops)}."""

    def __init__(self):
        self.offloads = {}

    def visit_FunctionDef(self, node):
        is_kernel = any(isinstance(d, ast.Attribute) and d.attr == "kernel" and isinstance(d.value, ast.Name) and
                        d.value.id == "ti" for d in node.decorator_list)
        if not is_kernel:
            return node
        body, k, block_loop, plain = [], 0, None, 0
        for stmt in node.body:
            if isinstance(stmt, ast.For):
                if "simt.block" in ast.unparse(stmt):
                    block_loop = k
                elif isinstance(stmt.iter, ast.Call) and isinstance(stmt.iter.func, ast.Name) and stmt.iter.func.id == "range":
                    # a parallel loop over a plain range: its iterations can be dealt to worker processes
                    stmt.iter = ast.copy_location(ast.Call(func=ast.Name(id="__ti_prange__", ctx=ast.Load()),
                                                           args=[stmt.iter], keywords=[]), stmt.iter)
                    plain += 1
                guard = ast.Call(func=ast.Name(id="__ti_offload__", ctx=ast.Load()), args=[ast.Constant(value=k)],
                                 keywords=[])
                stmt = ast.copy_location(ast.If(test=guard, body=[stmt], orelse=[]), stmt)
                k += 1
            body.append(stmt)
        node.body = body
        self.offloads[node.name] = (k, block_loop, plain)   # (top-level loops, index of the block loop, plain-range loops)
        return node


def load_reference(reference_root: str, modules=("Camera", "utils", "SphericalHarmonics", "GaussianPoint3D",
                                                  "GaussianPointCloudRasterisation"), source_patches=None):
    """Execute the reference's modules, from where they lie, against the emulated ``taichi``; returns
    {module name: module}.  ``dataclass_wizard`` (absent here too) is replaced by an empty ``YAMLWizard``.
    source_patches: {module name: [(old, new), ...]} literal one-for-one text replacements applied to the source
    before execution (each ``old`` must occur exactly once) -- used for exactly one experiment, the stable-sort patch of
    make_reference_operator_vectors.py; vectors generated with a patch say so."""
    ti, tm = build_taichi_module()
    sys.modules["taichi"], sys.modules["taichi.math"] = ti, tm
    wizard = types.ModuleType("dataclass_wizard")
    wizard.YAMLWizard = type("YAMLWizard", (), {})
    sys.modules["dataclass_wizard"] = wizard
    pkg_name = "taichi_3d_gaussian_splatting"
    pkg_dir = os.path.join(reference_root, pkg_name)
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [pkg_dir]
    sys.modules[pkg_name] = pkg
    loaded = {}
    for name in modules:
        path = os.path.join(pkg_dir, name + ".py")
        with open(path) as fh:
            source = fh.read()
        for old, new in (source_patches or {}).get(name, []):
            assert source.count(old) == 1, (name, old)
            source = source.replace(old, new)
        offloads = _OffloadRewriter()
        tree = _AtomicRewriter().visit(ast.parse(source, filename=path))
        if EMU_FMA or EMU_RCP_DIV:
            tree = _ArithmeticRewriter(EMU_FMA, EMU_RCP_DIV).visit(tree)
        tree = ast.fix_missing_locations(offloads.visit(tree))
        mod = types.ModuleType(f"{pkg_name}.{name}")
        mod.__file__, mod.__package__ = path, pkg_name
        mod.__dict__["__ti_atomic_add__"] = ti_atomic_add
        mod.__dict__["__ti_fma__"], mod.__dict__["__ti_neg__"], mod.__dict__["__ti_div__"] = ti_fma, ti_neg, ti_div
        mod.__dict__["__ti_offload__"], mod.__dict__["__ti_offloads__"] = _offload, offloads.offloads
        mod.__dict__["__ti_prange__"] = _prange
        sys.modules[mod.__name__] = mod
        import linecache
        linecache.cache[path] = (len(source), None, source.splitlines(True), path)   # inspect.getsource for kernels
        exec(compile(tree, path, "exec"), mod.__dict__)
        setattr(pkg, name, mod)
        loaded[name] = mod
    return loaded
