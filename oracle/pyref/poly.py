"""Oracle restatement of kateth `src/kzg/poly.rs`."""
from . import bls
from .bls import R


def evaluate(coeffs, point: int, setup) -> int:
    """`Polynomial::evaluate` (src/kzg/poly.rs:10-33)."""
    roots = setup.roots_of_unity_brp
    n = len(coeffs)
    for i in range(n):  # :14-18
        if point == roots[i]:
            return coeffs[i]
    ev = 0
    for i in range(n):  # :23-28 (one field division per term)
        numer = coeffs[i] * roots[i] % R
        denom = (point - roots[i]) % R
        ev = (ev + bls.fr_div(numer, denom)) % R
    # :31-32 ; n >= 2 so Fr::pow's power==0 quirk (Q2) is not reached here
    term = bls.fr_div((bls.fr_pow_reference(point, n) - 1) % R, n % R)
    return ev * term % R


def prove(coeffs, point: int, setup):
    """`Polynomial::prove` (src/kzg/poly.rs:36-71) -> (eval, proof point)."""
    roots = setup.roots_of_unity_brp
    n = len(coeffs)
    ev = evaluate(coeffs, point, setup)
    quotient = []
    for i in range(n):
        numer = (coeffs[i] - ev) % R
        denom = (roots[i] - point) % R
        if denom != 0:
            q = bls.fr_div(numer, denom)
        else:  # :50-64
            q = 0
            for j in range(n):
                if j == i:
                    continue
                coefficient = (coeffs[j] - ev) % R
                nm = coefficient * roots[j] % R
                dn = (roots[i] * roots[i] - roots[i] * roots[j]) % R
                q = (q + bls.fr_div(nm, dn)) % R
        quotient.append(q)
    return ev, bls.g1_lincomb_pippenger(setup.g1_lagrange_brp, quotient)
