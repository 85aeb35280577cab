"""QV block of src/jasper.sh:235-257 (awk column sums + GNU bc fixed-scale arithmetic).

PARITY UNPINNED for the printed digits: GNU bc is a third-party tool that is not part of the reference tree and is
not installed in the build container.  The inputs (bad, total) ARE pinned by the oracle.  The emulation follows
bc's documented rules: `scale=N` truncates (toward zero) quotients and the results of l()/e() to N decimals, a
product keeps min(a+b, max(scale, a, b)) decimals, and numbers in (-1,1) print without a leading zero.
Sums are exact integers (gawk behaviour; mawk would print %.6g above 2^31).
"""
from decimal import Decimal, ROUND_DOWN, getcontext

getcontext().prec = 120


def _trunc(x, scale):
    q = Decimal(1).scaleb(-scale)
    return x.quantize(q, rounding=ROUND_DOWN)


def _bc_str(x):
    """bc prints at the value's own scale, without a leading zero for |x| < 1"""
    s = format(x, "f")
    if s.startswith("0."):
        s = s[1:]
    elif s.startswith("-0."):
        s = "-" + s[2:]
    return s


def q_value(bad, total, kmer):
    """returns the string jasper.sh logs after 'Q value = ' (src/jasper.sh:239-246 / 249-256)"""
    if int(total) == 0:
        # bc: "Divide by zero" -> empty $pgood -> the later bc calls print errors, the (( ... )) test fails -> else branch
        return "Inf"
    bad = Decimal(int(bad))
    total = Decimal(int(total))
    pgood = Decimal(1) - _trunc(bad / total, 10)                      # scale=10; 1-bad/total
    if pgood <= 0:
        # l() of a non-positive number: bc returns a huge negative constant; e() of that underflows to 0 -> rate 1
        error_rate = Decimal(1)
    else:
        inv_k = _trunc(Decimal(1) / Decimal(int(kmer)), 50)          # (1/K) at scale=50
        lg = _trunc(pgood.ln(), 50)                                   # l(pgood)
        prod = _trunc(lg * inv_k, 50)                                 # product scale = min(100, 50)
        error_rate = Decimal(1) - _trunc(prod.exp(), 50)              # 1 - e(...)
    if error_rate > 0:                                                # (( $(echo "$err > 0" | bc -l) ))
        l_err = _trunc(error_rate.ln(), 5)                            # scale=5
        num = _trunc(Decimal(-10) * l_err, 5)
        l10 = _trunc(Decimal(10).ln(), 5)
        return _bc_str(_trunc(num / l10, 5))
    return "Inf"
