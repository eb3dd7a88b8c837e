"""The algebra of the verification path's evaluation kernel (kateth_amd/csrc/verify_kernels.cuh, k_eval_frac), restated with
Python integers and checked against the oracle's Polynomial::evaluate (src/kzg/poly.rs:10-33) on the CPU:

  * in bit-reversed order elements 8o .. 8o+7 sit at w, -w, iw, -iw, cw, -cw, icw, -icw (i, c: primitive 4th / 8th roots);
  * pair  e0 x/(z-x) - e1 x/(z+x) = x u / (z^2 - x^2),  u = (e0 - e1) z + (e0 + e1) x;
  * quad  x [u d' + u' (i d)] / (z^4 - x^4),  d = z^2 - x^2, d' = z^2 + x^2;  the second quad (x = cw) reuses w^2 and i w^2;
  * oct   w [A dd' + A' (c dd)] / (z^8 - w^8),  dd = z^4 - w^4, dd' = z^4 + w^4;
  * lane fraction (N, D) <- (N ddd + B D, D ddd); the merged D is z^4096 - 1, so y = N / 4096: no inversion anywhere.

The kernel computes exactly these products (18 per oct, with 10 Montgomery reductions); this test pins the identities, the
GPU tests pin the kernel (tests/test_gpu_parity.py: test_evaluation_kernel_on_and_off_the_domain and friends)."""
import random
import types

from oracle.pyref import domain, poly
from oracle.pyref.bls import R


def oct_fraction_evaluate(elements, z, roots):
    i4, c8 = roots[2], roots[4]  # w = 1 for oct 0: roots[2] = i, roots[4] = c
    assert i4 * i4 % R == R - 1 and c8 * c8 % R == i4
    z2, z4, z8 = z * z % R, pow(z, 4, R), pow(z, 8, R)
    iz2, cz4 = i4 * z2 % R, c8 * z4 % R
    N, D = 0, 1
    products = 0
    for o in range(512):
        e = elements[8 * o:8 * o + 8]
        w = roots[8 * o]
        assert [roots[8 * o + k] for k in range(8)] == [w, R - w, i4 * w % R, R - i4 * w % R, c8 * w % R, R - c8 * w % R,
                                                        i4 * c8 * w % R, R - i4 * c8 * w % R]
        w2, w4, w8 = w * w % R, pow(w, 4, R), pow(w, 8, R)
        iw2, cw4 = i4 * w2 % R, c8 * w4 % R

        def pair(e0, e1, x):
            return ((e0 - e1) * z + (e0 + e1) * x) % R  # two products, one reduction

        u0, u1 = pair(e[0], e[1], w), pair(e[2], e[3], i4 * w % R)
        u2, u3 = pair(e[4], e[5], c8 * w % R), pair(e[6], e[7], i4 * c8 * w % R)
        a1 = (u0 * (z2 + w2) + u1 * (iz2 - iw2)) % R   # quad 1: d' = z^2 + w^2, i d = i z^2 - i w^2
        a2 = (u2 * (z2 + iw2) + u3 * (iz2 + w2)) % R   # quad 2: (cw)^2 = i w^2, so d' = z^2 + i w^2, i d = i z^2 + w^2
        b = (a1 * (z4 + w4) + a2 * (cz4 - cw4)) % R    # dd' = z^4 + w^4, c dd = c z^4 - c w^4
        b = b * w % R                                  # the root, once per oct
        ddd = (z8 - w8) % R
        N = (N * ddd + b * D) % R
        D = D * ddd % R
        products += 4 * 2 + 2 * 2 + 2 + 1 + 2 + 1
    assert products == 18 * 512
    assert D == (pow(z, 4096, R) - 1) % R  # the merged denominator cancels the barycentric factor
    return N * pow(4096, R - 2, R) % R


def test_oct_fraction_sum_is_the_barycentric_evaluation():
    rng = random.Random(0x0C7)
    roots = domain.bit_reversal_permutation(domain.roots_of_unity(4096))
    setup = types.SimpleNamespace(roots_of_unity_brp=roots)
    elements = [rng.randrange(R) for _ in range(4096)]
    for z in (0, 1, 2, R - 1, rng.randrange(R), rng.randrange(R)):
        if z in (1, R - 1):
            continue  # on the domain: the kernel returns the element itself (poly.rs:14-18), no fraction sum
        assert oct_fraction_evaluate(elements, z, roots) == poly.evaluate(elements, z, setup), hex(z)
