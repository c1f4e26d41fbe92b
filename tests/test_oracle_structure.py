"""Known-answer and structural tests of the oracle's restatement (SURVEY.md section 4 lists what
the reference lacks): CSR offsets vs a brute-force count, the symmetry identities the
matrix-free HIP operator relies on, SpMV vs a dense build, PCG vs a dense solve."""
import ctypes as C

import numpy as np
import pytest

from octane_amd import synth


def _assemble(oracle, nx, ny, nc=1, al1=0.5, seed=3, u=None, v=None, lambdac=0.0):
    """Drive oct_oracle_assemble directly on a level built from a lattice scene."""
    L = oracle.lib()
    a, b = synth.lattice_scene(nx, ny, seed=seed, nchan=nc)
    F = np.float32
    gx1, gy1, gx2, gy2, gxx, gxy, gyy, dead = (np.zeros((nc, ny, nx), F) for _ in range(8))
    L.oct_oracle_gradient(a, gx1, gy1, nx, ny, nc)
    L.oct_oracle_gradient(b, gx2, gy2, nx, ny, nc)
    L.oct_oracle_gradient(gx2, gxx, dead, nx, ny, nc)
    L.oct_oracle_gradient(gy2, gxy, gyy, nx, ny, nc)
    rng = np.random.RandomState(seed)
    u = (2.5 + 0.3 * rng.randn(ny, nx)).astype(F) if u is None else u
    v = (-1.0 + 0.3 * rng.randn(ny, nx)).astype(F) if v is None else v
    n = nx * ny
    nnz = 12 * n - 4 * nx - 4 * ny

    class Level(C.Structure):
        _fields_ = [("xi", C.c_int), ("yi", C.c_int), ("nc", C.c_int)] + [(k, C.c_void_p) for k in
                    ("img1", "img2", "gx1", "gy1", "gx2", "gy2", "gxx", "gxy", "gyy")]

    class System(C.Structure):
        _fields_ = [("nrows", C.c_int), ("nnz", C.c_long), ("val", C.c_void_p), ("row", C.c_void_p),
                    ("col", C.c_void_p), ("rowptr", C.c_void_p), ("diag", C.c_void_p), ("rhs", C.c_void_p)]

    class Planes(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("a1", "a2", "a4", "a5", "a6", "a7", "a8", "bu", "bv")]

    keep = dict(a=a, b=b, gx1=gx1, gy1=gy1, gx2=gx2, gy2=gy2, gxx=gxx, gxy=gxy, gyy=gyy)
    lev = Level(nx, ny, nc, *[keep[k].ctypes.data for k in ("a", "b", "gx1", "gy1", "gx2", "gy2", "gxx", "gxy", "gyy")])
    val = np.full(nnz, np.nan, F)
    row = np.full(nnz, -1, np.int32)
    col = np.full(nnz, -1, np.int32)
    rowptr = np.full(2 * n, -1, np.int32)
    diag = np.zeros(2 * n, F)
    rhs = np.zeros(2 * n, F)
    S = System(2 * n, nnz, val.ctypes.data, row.ctypes.data, col.ctypes.data, rowptr.ctypes.data, diag.ctypes.data, rhs.ctypes.data)
    planes = np.zeros((9, ny, nx), F)
    P = Planes(*[planes[i].ctypes.data for i in range(9)])
    ut = np.zeros((ny, nx), F)
    vt = np.zeros((ny, nx), F)
    L.oct_oracle_assemble.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    L.oct_oracle_assemble(C.byref(lev), u.ctypes.data, v.ctypes.data, ut.ctypes.data, vt.ctypes.data,
                          al1, 5.0, 0.2, lambdac, 1, C.byref(S), C.byref(P))
    return dict(val=val, row=row, col=col, rowptr=rowptr, diag=diag, rhs=rhs, planes=planes, nnz=nnz, n=n, keep=keep)


@pytest.mark.parametrize("nx,ny", [(2, 2), (3, 5), (7, 4), (16, 9)])
def test_csr_offsets_equal_bruteforce_count(oracle, nx, ny):
    """ref .cu:868-913's closed form == count of entries in all preceding rows."""
    L = oracle.lib()
    per_pixel = lambda i, j: 2 * ((j > 0) + (i > 0) + 2 + (i < nx - 1) + (j < ny - 1))
    running = 0
    for n in range(nx * ny):
        i, j = n % nx, n // nx
        assert L.oct_oracle_nnz_before(n, i, j, nx, ny) == running
        running += per_pixel(i, j)
    assert running == 12 * nx * ny - 4 * nx - 4 * ny     # An, ref .cu:600


@pytest.mark.parametrize("nx,ny,nc", [(9, 7, 1), (12, 5, 2), (2, 2, 1), (5, 3, 3)])
def test_csr_is_fully_and_consistently_filled(oracle, nx, ny, nc):
    r = _assemble(oracle, nx, ny, nc)
    assert not np.isnan(r["val"]).any() and (r["col"] >= 0).all() and (r["rowptr"] >= 0).all()
    assert (np.diff(r["rowptr"]) >= 4).all() and (np.diff(r["rowptr"]) <= 6).all()
    # each stored entry's row tag agrees with the row-pointer structure
    ends = np.append(r["rowptr"][1:], r["nnz"])
    for k in range(2 * r["n"]):
        assert (r["row"][r["rowptr"][k]:ends[k]] == k).all()


@pytest.mark.parametrize("nx,ny,al1", [(11, 8, 1.0), (11, 8, 0.5), (11, 8, 0.0), (2, 3, 0.0), (6, 2, 0.5)])
def test_neighbour_weight_identities_hold_bitwise(oracle, nx, ny, al1):
    """The HIP operator stores only a7 (east) and a8 (north): it needs
    a5(i,j) == a7(i-1,j), a6(i,j) == a8(i,j-1), and at the mirrored borders a5(0,j) == a7(0,j),
    a7(nx-1,j) == a5(nx-1,j), a6(i,0) == a8(i,0), a8(i,ny-1) == a6(i,ny-1) -- bit for bit."""
    p = _assemble(oracle, nx, ny, al1=al1)["planes"]
    a5, a6, a7, a8 = p[3], p[4], p[5], p[6]
    assert np.array_equal(a5[:, 1:], a7[:, :-1])
    assert np.array_equal(a6[1:, :], a8[:-1, :])
    assert np.array_equal(a5[:, 0], a7[:, 0])
    assert np.array_equal(a7[:, -1], a5[:, -1])
    assert np.array_equal(a6[0, :], a8[0, :])
    assert np.array_equal(a8[-1, :], a6[-1, :])


def _dense_from_planes(p, nx, ny):
    """Dense 2N x 2N matrix the way the HIP pass applies it (merged border weights)."""
    a1, a2, a4, a7, a8 = p[0], p[1], p[2], p[5], p[6]
    n = nx * ny
    A = np.zeros((2 * n, 2 * n), np.float64)
    for j in range(ny):
        for i in range(nx):
            k = i + nx * j
            wS = wW = wE = wN = None
            if j > 0:
                wS = np.float32(a8[j - 1, i]) + (np.float32(a8[j, i]) if j == ny - 1 else np.float32(0))
            if i > 0:
                wW = np.float32(a7[j, i - 1]) + (np.float32(a7[j, i]) if i == nx - 1 else np.float32(0))
            if i < nx - 1:
                wE = np.float32(a7[j, i]) * (2 if i == 0 else 1)
            if j < ny - 1:
                wN = np.float32(a8[j, i]) * (2 if j == 0 else 1)
            for comp in range(2):
                r = 2 * k + comp
                if wS is not None: A[r, 2 * (k - nx) + comp] = wS
                if wW is not None: A[r, 2 * (k - 1) + comp] = wW
                if wE is not None: A[r, 2 * (k + 1) + comp] = wE
                if wN is not None: A[r, 2 * (k + nx) + comp] = wN
            A[2 * k, 2 * k] = a1[j, i]; A[2 * k, 2 * k + 1] = a2[j, i]
            A[2 * k + 1, 2 * k] = a2[j, i]; A[2 * k + 1, 2 * k + 1] = a4[j, i]
    return A


@pytest.mark.parametrize("nx,ny", [(8, 8), (5, 7), (2, 2)])
def test_matrix_free_operator_equals_reference_csr(oracle, nx, ny):
    r = _assemble(oracle, nx, ny, al1=0.5)
    n2 = 2 * r["n"]
    ends = np.append(r["rowptr"][1:], r["nnz"])
    A_csr = np.zeros((n2, n2), np.float64)
    for k in range(n2):
        for e in range(r["rowptr"][k], ends[k]):
            A_csr[k, r["col"][e]] += r["val"][e]
    A_mf = _dense_from_planes(r["planes"], nx, ny)
    assert np.array_equal(A_csr.astype(np.float32), A_mf.astype(np.float32))
    # and oct_oracle_spmv applies exactly that matrix
    x = np.random.RandomState(0).randn(n2).astype(np.float32)
    y = np.zeros(n2, np.float32)
    oracle.lib().oct_oracle_spmv.argtypes = [C.c_void_p] * 4 + [C.c_long, C.c_int, C.c_void_p]
    oracle.lib().oct_oracle_spmv(r["val"].ctypes.data, r["rowptr"].ctypes.data, r["col"].ctypes.data, x.ctypes.data, r["nnz"], n2, y.ctypes.data)
    np.testing.assert_allclose(y, A_csr @ x.astype(np.float64), rtol=2e-5, atol=2e-5)


def test_pcg_converges_to_dense_solve_on_8x8(oracle):
    """With enough iterations the Jacobi-PCG of ref .cu:1105-1182 reaches the dense solution."""
    nx = ny = 8
    r = _assemble(oracle, nx, ny, al1=1.0)
    n2 = 2 * r["n"]

    class System(C.Structure):
        _fields_ = [("nrows", C.c_int), ("nnz", C.c_long), ("val", C.c_void_p), ("row", C.c_void_p),
                    ("col", C.c_void_p), ("rowptr", C.c_void_p), ("diag", C.c_void_p), ("rhs", C.c_void_p)]

    class Work(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("z", "p", "rk", "tmp", "ident")]

    A = _dense_from_planes(r["planes"], nx, ny)
    b = r["rhs"].astype(np.float64).copy()
    want = np.linalg.solve(A, b)
    bufs = [np.zeros(n2, np.float32) for _ in range(4)] + [np.zeros(n2, np.int32)]
    W = Work(*[x.ctypes.data for x in bufs])
    S = System(n2, r["nnz"], r["val"].ctypes.data, r["row"].ctypes.data, r["col"].ctypes.data,
               r["rowptr"].ctypes.data, r["diag"].ctypes.data, r["rhs"].ctypes.data)
    x = np.zeros(n2, np.float32)
    L = oracle.lib()
    L.oct_oracle_pcg.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    L.oct_oracle_pcg.restype = C.c_int
    its = L.oct_oracle_pcg(C.byref(S), x.ctypes.data, np.float32(1e-14), 400, C.byref(W))
    assert 0 < its <= 400
    assert np.linalg.norm(x - want) / np.linalg.norm(want) < 2e-4


def test_gradient_known_answers(oracle):
    """4th-order stencil is exact on cubics away from the clamped border; clamp (not reflect) at it."""
    nx, ny = 12, 9
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    f = (0.5 * i ** 3 - 2 * i ** 2 + 3 * j ** 2 + j).astype(np.float32)[None]
    gx = np.zeros_like(f); gy = np.zeros_like(f)
    oracle.lib().oct_oracle_gradient(f, gx, gy, nx, ny, 1)
    np.testing.assert_allclose(gx[0, :, 2:-2], (1.5 * i ** 2 - 4 * i)[:, 2:-2], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(gy[0, 2:-2, :], (6 * j + 1)[2:-2, :], rtol=1e-5, atol=1e-3)
    # border column 0 uses f[-1]=f[-2]=f[0]
    want0 = (-f[0, :, 2] + 8. * f[0, :, 1] - 8. * f[0, :, 0] + f[0, :, 0]) / 12.0
    np.testing.assert_allclose(gx[0, :, 0], want0, rtol=1e-6)


def test_multichannel_decimation_samples_channel_zero(oracle):
    """Quirk kept from ref .cu:406: every channel of a decimated level is channel 0."""
    nx, ny = 20, 16
    img = np.random.RandomState(2).rand(2, ny, nx).astype(np.float32)
    out = np.zeros((2, 8, 10), np.float32)
    oracle.lib().oct_oracle_decimate(img, out, nx, ny, 2, 0.5)
    assert np.array_equal(out[0], out[1])
    assert np.array_equal(out[0], img[0, ::2, ::2])


def test_openmp_flavour_is_bit_identical_to_the_scalar_one(oracle):
    """The multi-core build (bench.py's cpu_baseline) splits only order-independent loops and adds the block sums
    of the launch-geometry dot product in block order: same bits as the scalar build, two channels included."""
    a, b = synth.lattice_scene(97, 61, seed=5, nchan=2)
    prm = oracle.FlowParams(kiters=3, liters=2, cgiters=12, lambdac=0.3)
    us, vs, its = oracle.flow(a, b, prm, dot_threads=oracle.REF_GRID_THREADS)
    uo, vo, ito = oracle.flow(a, b, prm, flavour="omp", dot_threads=oracle.REF_GRID_THREADS)
    assert its == ito
    assert np.array_equal(us, uo) and np.array_equal(vs, vo)
    assert oracle.num_threads("strict") == 1 and oracle.num_threads("omp") >= 1
