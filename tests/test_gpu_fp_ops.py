"""FixedPointInstructions one operation at a time (SURVEY 8 a4 – a17; src/gadget/fixed_point.rs:213-460): the cells, lookup cells, gate
flags and result of every operation the chips are composed of, on the GPU (vdb_wit_fp_op: one call per lane) against the oracle's
`orc_fp_op`, bit for bit — random operands of both signs, zero, ones, the smallest and the largest magnitudes the chip accepts; results
against f64 at the reference tests' 1e-6 where the operation has an f64 counterpart."""
import math
import zlib

import numpy as np
import pytest

from test_gpu_witness import assert_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from halo2_vectordb_amd import api as a
    a.init()
    return a


UNARY = ("neg", "qabs", "is_neg", "qsqrt", "qlog2", "qexp2", "qlog", "qexp", "signed_div_scale", "sign", "clip", "qsin", "qcos", "qtan", "qsinh", "qcosh", "qtanh")
BINARY = ("qadd", "qsub", "qmul", "qdiv", "qmin", "qmax", "qpow", "bit_xor", "cond_neg", "qmod")
F64 = dict(qadd=lambda x, y: x + y, qsub=lambda x, y: x - y, qmul=lambda x, y: x * y, qdiv=lambda x, y: x / y, neg=lambda x: -x, qabs=abs,
           qmin=min, qmax=max, qsqrt=math.sqrt, qlog2=math.log2, qexp2=lambda x: 2.0 ** x, qlog=math.log, qexp=math.exp, qpow=lambda x, y: x ** y,
           qsin=math.sin, qcos=math.cos, qtan=math.tan, qsinh=math.sinh, qcosh=math.cosh, qtanh=math.tanh, clip=lambda x: x, qmod=lambda x, y: x % y)


def _operands(rng, op, n):
    """values inside the operation's domain"""
    if op in ("qsqrt", "qlog2", "qlog"):
        x = np.concatenate([rng.uniform(0.01, 200.0, n - 3), [1.0, 0.5, 4.0]])
        return x, None
    if op in ("qexp2", "qexp"):
        x = np.concatenate([rng.uniform(-8.0, 8.0, n - 3), [0.0, 1.0, -1.0]])
        return x, None
    if op == "qpow":
        return rng.uniform(0.1, 6.0, n), rng.uniform(-2.0, 2.0, n)
    if op in ("qsin", "qcos", "qsinh", "qcosh", "qtanh"):
        return np.concatenate([rng.uniform(-9.0, 9.0, n - 3), [0.25 * math.pi, 1.0, -3.5]]), None
    if op == "qtan":                             # away from the poles
        x = rng.uniform(-9.0, 9.0, n)
        x[np.abs(np.cos(x)) < 0.05] = 0.3
        return x, None
    if op == "qmod":
        return rng.uniform(-50.0, 50.0, n), rng.uniform(0.1, 7.0, n)
    if op == "bit_xor":
        return rng.integers(0, 2, n).astype(np.float64), rng.integers(0, 2, n).astype(np.float64)
    x = np.concatenate([rng.uniform(-300.0, 300.0, n - 4), [0.0, 1.0, -1.0, 2.0 ** -40]])
    y = np.concatenate([rng.uniform(-300.0, 300.0, n - 4), [1.0, -1.0, 3.0, -2.0 ** -20]])
    if op == "cond_neg":
        y = rng.integers(0, 2, n).astype(np.float64)
    if op == "qdiv":
        y[np.abs(y) < 1e-3] = 0.5
    return x, (None if op in UNARY else y)


@pytest.mark.parametrize("op", UNARY + BINARY)
@pytest.mark.parametrize("P,L", [(48, 13), (32, 9)])
def test_every_operation_against_the_oracle(api, O, op, P, L):
    rng = np.random.default_rng([zlib.crc32(op.encode()), P])          # (not hash(): that one differs from process to process)
    n = 70
    x, y = _operands(rng, op, n)
    qa = O.quantize(x, P)
    raw_bits = op in ("bit_xor", "cond_neg")                    # flags are field bits 0 / 1, not quantized numbers
    qb = None if y is None else (O.fr_from_ints([int(v) for v in y]) if raw_bits else O.quantize(y, P))
    if op == "bit_xor":
        qa = O.fr_from_ints([int(v) for v in x])
    if op == "signed_div_scale":                                # its operand is a product: twice the scale
        qa = O.fr_mul(qa, O.quantize(rng.uniform(-3, 3, n), P))
    c = O.Ctx(store=True, keygen=True)
    want = np.stack([c.op(op, qa[i], None if qb is None else qb[i], P=P, L=L) for i in range(n)])
    assert c.err == 0
    got = api.wit_fp_op(op, qa, qb, P=P, L=L, selectors=True)
    assert np.array_equal(got["result"], want), op
    assert_streams(got, c)
    assert got["stream"].shape[0] % n == 0 and c.check_gates(L) == 0
    if op in F64 and P == 48:
        deq = O.dequantize(got["result"], P)
        for i in range(n):
            f = F64[op](float(x[i]), float(y[i])) if y is not None else F64[op](float(x[i]))
            if 0 < abs(f) < 2.0 ** -40:
                continue       # below the chip's resolution; a result of -1 ulp meets dequantization's abs(v) - 2 quirk
            assert abs(float(deq[i]) - f) <= 1e-6 * max(abs(f), 1.0), (op, x[i], None if y is None else y[i], float(deq[i]), f)


def test_division_by_zero_is_refused_and_sizes_are_per_call(api, O):
    q = O.quantize(np.array([1.5, 2.5]))
    z = O.quantize(np.array([0.25, 0.0]))
    with pytest.raises(api.VdbError) as e:
        api.wit_fp_op("qdiv", q, z)
    assert e.value.code == -5
    one = api.wit_fp_op("qmul", q[:1], z[:1])
    two = api.wit_fp_op("qmul", q, z)
    assert two["stream"].shape[0] == 2 * one["stream"].shape[0] and two["lookup"].shape[0] == 2 * one["lookup"].shape[0]
    assert np.array_equal(two["stream"][: one["stream"].shape[0]], one["stream"])
    empty = api.wit_fp_op("qadd", np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))
    assert empty["stream"].shape[0] == 0 and empty["result"].shape[0] == 0
