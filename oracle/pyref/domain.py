"""Oracle restatement of kateth `src/math.rs` (roots of unity, bit reversal)."""
from .bls import R

PRIMITIVE_ROOT_OF_UNITY = 7  # src/math.rs:5


def primitive_root_of_unity(order: int) -> int:
    """src/math.rs:7-14.  `Fr::MAX / order` is a *field* division of r-1 by
    `order`; it equals the integer quotient because order | r-1."""
    power = (R - 1) * pow(order, -1, R) % R
    assert power == (R - 1) // order
    return pow(PRIMITIVE_ROOT_OF_UNITY, power, R)


def roots_of_unity(order: int):
    """src/math.rs:16-29: [1, w, w^2, ...]."""
    w = primitive_root_of_unity(order)
    out, cur = [], 1
    for _ in range(order):
        out.append(cur)
        cur = cur * w % R
    return out


def bit_reversal_permutation_index(index: int, length: int) -> int:
    """src/math.rs:72-74."""
    bits = length.bit_length() - 1
    return int(format(index, "0{}b".format(bits))[::-1], 2) if bits else 0


def bit_reversal_permutation(elements):
    """src/math.rs:34-46: panics (here: AssertionError) unless len is a power of two."""
    n = len(elements)
    assert n and (n & (n - 1)) == 0
    return [elements[bit_reversal_permutation_index(i, n)] for i in range(n)]
