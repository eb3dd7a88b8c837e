"""The algebra of the verification path's evaluation kernel (kateth_amd/csrc/verify_kernels.cuh, k_eval_frac), restated with
Python integers and checked against the oracle's Polynomial::evaluate (src/kzg/poly.rs:10-33) on the CPU:

  * in bit-reversed order elements 16h + 2k, 2k + 1 sit at +-w rho_k, rho = 1, i, c, ic, s, is, cs, ics (i, c, s: the primitive
    4th / 8th / 16th roots of unity);
  * pair  e0 x/(z-x) - e1 x/(z+x) = x u / (z^2 - x^2),  u = (e0 - e1) z + (e0 + e1) x;
  * quad  x [u d' + u' (i d)] / (z^4 - x^4),  d = z^2 - x^2, d' = z^2 + x^2;  the second quad of an oct (at cx) reuses x^2, i x^2;
  * oct   x [A dd' + A' (c dd)] / (z^8 - x^8),  dd = z^4 - x^4, dd' = z^4 + x^4;  the second oct (x = s w) has the same shape
          on (x^2, i x^2, x^4, c x^4) = (c w^2, i c w^2, i w^4, c i w^4);
  * hex   w [B ddd' + B' (s ddd)] / (z^16 - w^16),  ddd = z^8 - w^8, ddd' = z^8 + w^8;
  * lane fraction (N, D) <- (N dddd + H D, D dddd); the merged D is z^4096 - 1, so y = N / 4096: no inversion anywhere.

The kernel computes exactly these products (34 per hex, with 18 Montgomery reductions); this test pins the identities, the
GPU tests pin the kernel (tests/test_gpu_parity.py: test_evaluation_kernel_on_and_off_the_domain and friends)."""
import random
import types

from oracle.pyref import domain, poly
from oracle.pyref.bls import R


def hex_fraction_evaluate(elements, z, roots):
    i4, c8, s16 = roots[2], roots[4], roots[8]  # w = 1 for hex 0
    assert i4 * i4 % R == R - 1 and c8 * c8 % R == i4 and s16 * s16 % R == c8
    z2, z4, z8, z16 = z * z % R, pow(z, 4, R), pow(z, 8, R), pow(z, 16, R)
    iz2, cz4, sz8 = i4 * z2 % R, c8 * z4 % R, s16 * z8 % R
    N, D = 0, 1
    products = 0
    rho = [1, i4, c8, i4 * c8 % R, s16, i4 * s16 % R, c8 * s16 % R, i4 * c8 * s16 % R]
    for h in range(256):
        e = elements[16 * h:16 * h + 16]
        w = roots[16 * h]
        for k in range(8):
            assert roots[16 * h + 2 * k] == w * rho[k] % R and roots[16 * h + 2 * k + 1] == R - w * rho[k] % R
        w2, w4, w8, w16 = w * w % R, pow(w, 4, R), pow(w, 8, R), pow(w, 16, R)
        # the slots of the two octs: (x^2, i x^2, x^4, c x^4)
        slots = [(w2, i4 * w2 % R, w4, c8 * w4 % R), (c8 * w2 % R, i4 * c8 * w2 % R, i4 * w4 % R, c8 * i4 * w4 % R)]

        def pair(e0, e1, x):
            return ((e0 - e1) * z + (e0 + e1) * x) % R  # two products, one reduction

        bt = []
        for oc in range(2):
            x2, ix2, x4, cx4 = slots[oc]
            x = [w * rho[4 * oc + j] % R for j in range(4)]
            assert x[0] * x[0] % R == x2 and x[2] * x[2] % R == ix2 and pow(x[0], 4, R) == x4
            ee = e[8 * oc:8 * oc + 8]
            u = [pair(ee[2 * j], ee[2 * j + 1], x[j]) for j in range(4)]
            a1 = (u[0] * (z2 + x2) + u[1] * (iz2 - ix2)) % R   # d' = z^2 + x^2, i d = i z^2 - i x^2
            a2 = (u[2] * (z2 + ix2) + u[3] * (iz2 + x2)) % R   # (cx)^2 = i x^2
            bt.append((a1 * (z4 + x4) + a2 * (cz4 - cx4)) % R)  # dd' = z^4 + x^4, c dd = c z^4 - c x^4
            products += 4 * 2 + 2 * 2 + 2
        ht = (bt[0] * (z8 + w8) + bt[1] * (sz8 - s16 * w8)) % R  # ddd' = z^8 + w^8, s ddd = s z^8 - s w^8
        hh = ht * w % R                                            # the root, once per hex
        d16 = (z16 - w16) % R
        N = (N * d16 + hh * D) % R
        D = D * d16 % R
        products += 2 + 1 + 2 + 1
    assert products == 34 * 256
    assert D == (pow(z, 4096, R) - 1) % R  # the merged denominator cancels the barycentric factor
    return N * pow(4096, R - 2, R) % R


def test_hex_fraction_sum_is_the_barycentric_evaluation():
    rng = random.Random(0x0C7)
    roots = domain.bit_reversal_permutation(domain.roots_of_unity(4096))
    setup = types.SimpleNamespace(roots_of_unity_brp=roots)
    elements = [rng.randrange(R) for _ in range(4096)]
    for z in (0, 2, rng.randrange(R), rng.randrange(R)):  # off the domain (on it the kernel returns the element, poly.rs:14-18)
        assert hex_fraction_evaluate(elements, z, roots) == poly.evaluate(elements, z, setup), hex(z)
