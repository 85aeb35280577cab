"""QV block of src/jasper.sh:235-257 (awk column sums + GNU bc fixed-scale arithmetic).

The inputs (bad, total) are pinned by the oracle.  The printed digits come out of GNU bc, a third-party tool that is neither part
of the reference tree nor installed in the build container, so they are restated here from bc's published sources (GNU bc 1.07.1:
`bc/libmath.b` for l() and e(), `lib/number.c` for the scale rules of + - * / and for bc_sqrt) and NOT checked against a bc
binary: PARITY UNPINNED for the last printed digit, by construction identical wherever bc follows its own documented algorithm.

What is restated (`_Bc` below: a number is an integer n and a scale s, value n / 10^s, as in number.c):
  * a + b, a - b     scale max(sa, sb), exact
  * a * b            computed exactly (scale sa + sb), then cut to min(sa + sb, max(scale, sa, sb)) decimals
  * a / b            cut to `scale` decimals;  every cut truncates toward zero
  * sqrt(a)          Newton iteration of bc_sqrt on truncated quotients (start 1 or 10^(digits/2); working scale 3, or the
                     argument's own scale below 1, tripled up to rscale + 1 each time two iterates agree to one unit), result
                     cut to rscale = max(scale, sa)
  * l(x)  libmath.b  scale += 6; x brought into (.5, 2) by repeated sqrt (f doubles each time); a = (x-1)/(x+1);
                     v = a + a^3/3 + a^5/5 + ... until a term truncates to 0; result f*v cut to the caller's scale
  * e(x)  libmath.b  |x| halved until <= 1 (f times); scale = 6 + scale + .44*x; v = 1 + x + x^2/2! + ... until a term
                     truncates to 0; v squared f times; 1/v for negative x; cut to the caller's scale
  * printing         all `scale` decimals, no leading zero for |x| < 1
`q_value_exact` keeps round 2's formulation (correctly rounded ln / exp, then truncated like bc's results): the two agree in all
but a few of 10^5 cases in the fifth decimal (tests/test_host_logic.py counts them), which is the size of what is unpinned.
Sums are exact integers (gawk behaviour; mawk would print %.6g above 2^31).
"""
from decimal import Decimal, ROUND_DOWN, getcontext

getcontext().prec = 120


# ---- GNU bc numbers -------------------------------------------------------------------------------------------------
def _cut(n, s, to):
    """n / 10^s cut (toward zero) to `to` decimals"""
    if to >= s:
        return n * 10 ** (to - s), to
    q = 10 ** (s - to)
    return (abs(n) // q) * (1 if n >= 0 else -1), to


class _Bc:
    """the arithmetic of one bc process: `scale` is bc's global"""

    def __init__(self, scale=0):
        self.scale = scale

    @staticmethod
    def lit(text):
        text = text.strip()
        neg = text.startswith("-")
        if neg:
            text = text[1:]
        ip, _, fp = text.partition(".")
        n = int((ip or "0") + fp)
        return (-n if neg else n), len(fp)

    @staticmethod
    def add(a, b):
        s = max(a[1], b[1])
        return a[0] * 10 ** (s - a[1]) + b[0] * 10 ** (s - b[1]), s

    @staticmethod
    def neg(a):
        return -a[0], a[1]

    def sub(self, a, b):
        return self.add(a, self.neg(b))

    def mul(self, a, b, scale=None):
        scale = self.scale if scale is None else scale
        full = a[1] + b[1]
        return _cut(a[0] * b[0], full, min(full, max(scale, a[1], b[1])))

    def div(self, a, b, scale=None):
        scale = self.scale if scale is None else scale
        if b[0] == 0:
            raise ZeroDivisionError("bc: divide by zero")
        # a/b = (na * 10^sb) / (nb * 10^sa); wanted: floor(|.| * 10^scale)
        num = abs(a[0]) * 10 ** (b[1] + scale)
        den = abs(b[0]) * 10 ** a[1]
        q = num // den
        return (q if (a[0] >= 0) == (b[0] >= 0) else -q), scale

    @staticmethod
    def cmp(a, b):
        s = max(a[1], b[1])
        x, y = a[0] * 10 ** (s - a[1]), b[0] * 10 ** (s - b[1])
        return (x > y) - (x < y)

    @staticmethod
    def _near_zero(a, scale):
        """number.c bc_is_near_zero: the first `scale` decimals are zero, or only the last of them is 1"""
        n, _ = _cut(abs(a[0]), a[1], scale)
        return n <= 1

    def sqrt(self, a):
        one = (1, 0)
        if a[0] == 0:
            return 0, 0
        c1 = self.cmp(a, one)
        if c1 == 0:
            return one
        rscale = max(self.scale, a[1])
        half = (5, 1)
        if c1 < 0:
            guess, cscale = one, a[1]
        else:
            n_len = len(str(abs(a[0]) // 10 ** a[1]))          # digits before the decimal point
            e2 = self.mul((n_len, 0), half, 0)
            guess, cscale = (10 ** (e2[0] // 10 ** e2[1]), 0), 3
        while True:
            prev = guess
            guess = self.div(a, guess, cscale)
            guess = self.add(guess, prev)
            guess = self.mul(guess, half, cscale)
            diff = self.add(guess, self.neg(prev))
            diff = (diff[0] * 10 ** max(0, cscale + 1 - diff[1]), max(diff[1], cscale + 1))
            if self._near_zero(diff, cscale):
                if cscale < rscale + 1:
                    cscale = min(cscale * 3, rscale + 1)
                else:
                    break
        return self.div(guess, one, rscale)

    # ---- libmath.b ---------------------------------------------------------------------------------------------------
    def l(self, x):
        if x[0] <= 0:
            return self.div(self.sub((1, 0), (10 ** self.scale, 0)), (1, 0))
        z = self.scale
        self.scale = 6 + z
        f = (2, 0)
        two, half = (2, 0), (5, 1)
        while self.cmp(x, two) >= 0:
            f = self.mul(f, two)
            x = self.sqrt(x)
        while self.cmp(x, half) <= 0:
            f = self.mul(f, two)
            x = self.sqrt(x)
        n = self.div(self.sub(x, (1, 0)), self.add(x, (1, 0)))
        v = n
        m = self.mul(n, n)
        i = 3
        while True:
            n = self.mul(n, m)
            e = self.div(n, (i, 0))
            if e[0] == 0:
                v = self.mul(f, v)
                self.scale = z
                return self.div(v, (1, 0))
            v = self.add(v, e)
            i += 2

    def e(self, x):
        m = x[0] < 0
        if m:
            x = self.neg(x)
        z = self.scale
        n = self.add(self.add((6, 0), (z, 0)), self.mul((44, 2), x))
        self.scale = x[1] + 1
        f = 0
        while self.cmp(x, (1, 0)) > 0:
            f += 1
            x = self.div(x, (2, 0))
            self.scale += 1
        self.scale = abs(n[0]) // 10 ** n[1]
        v = self.add((1, 0), x)
        a = x
        d = 1
        i = 2
        while True:
            a = self.mul(a, x)
            d *= i
            t = self.div(a, (d, 0))
            if t[0] == 0:
                while f > 0:
                    v = self.mul(v, v)
                    f -= 1
                self.scale = z
                return self.div((1, 0), v) if m else self.div(v, (1, 0))
            v = self.add(v, t)
            i += 1

    @staticmethod
    def show(a):
        """bc prints all of a number's decimals and no leading zero for |x| < 1"""
        n, s = a
        if n == 0:
            return "0"
        digits = str(abs(n)).rjust(s + 1, "0")
        ip, fp = (digits[:-s], digits[-s:]) if s else (digits, "")
        if ip.strip("0") == "":
            ip = "" if fp else "0"
        out = ip + ("." + fp if fp else "")
        return ("-" if n < 0 else "") + out


def _error_rate(bad, total, kmer):
    """the two bc calls of src/jasper.sh:239-240; returns bc's printed error rate (a string), or None for 'Inf'"""
    if int(total) == 0:
        # bc: "Divide by zero" -> empty $pgood -> the later bc calls print errors, the (( ... )) test fails -> else branch
        return None
    bc = _Bc(10)                                                        # echo "scale=10; 1-$bad/$total" | bc
    pgood = bc.sub((1, 0), bc.div((int(bad), 0), (int(total), 0)))
    if pgood[0] <= 0:
        # l() of a non-positive number is bc's "minus infinity" (1 - 10^scale); e() of that / K asks for a scale of ~10^48 digits,
        # which no bc finishes: taken as the limit, error rate 1
        return "1"
    bc = _Bc(50)                                                        # echo "scale=50; 1 - e(l($pgood)*(1/$KMER))" | bc -l
    pg = _Bc.lit(_Bc.show(pgood))
    arg = bc.mul(bc.l(pg), bc.div((1, 0), (int(kmer), 0)))
    return _Bc.show(bc.sub((1, 0), bc.e(arg)))


def q_value(bad, total, kmer):
    """returns the string jasper.sh logs after 'Q value = ' (src/jasper.sh:239-246 / 249-256)"""
    rate = _error_rate(bad, total, kmer)
    if rate is None:
        return "Inf"
    r = _Bc.lit(rate)
    if r[0] <= 0:                                                       # (( $(echo "$err > 0" | bc -l) ))
        return "Inf"
    bc = _Bc(5)                                                         # echo "scale=5; -10*l($err) / l(10)" | bc -l
    num = bc.mul((-10, 0), bc.l(r))
    return _Bc.show(bc.div(num, bc.l((10, 0))))


# ---- round 2's formulation, kept as the independent restatement the tests compare with -------------------------------------
def _trunc(x, scale):
    q = Decimal(1).scaleb(-scale)
    return x.quantize(q, rounding=ROUND_DOWN)


def _bc_str(x):
    s = format(x, "f")
    if s.startswith("0."):
        s = s[1:]
    elif s.startswith("-0."):
        s = "-" + s[2:]
    return s


def q_value_exact(bad, total, kmer):
    """correctly rounded ln / exp truncated where bc truncates (no series, no internal scale bump)"""
    if int(total) == 0:
        return "Inf"
    bad = Decimal(int(bad))
    total = Decimal(int(total))
    pgood = Decimal(1) - _trunc(bad / total, 10)
    if pgood <= 0:
        error_rate = Decimal(1)
    else:
        inv_k = _trunc(Decimal(1) / Decimal(int(kmer)), 50)
        lg = _trunc(pgood.ln(), 50)
        prod = _trunc(lg * inv_k, 50)
        error_rate = Decimal(1) - _trunc(prod.exp(), 50)
    if error_rate > 0:
        l_err = _trunc(error_rate.ln(), 5)
        num = _trunc(Decimal(-10) * l_err, 5)
        l10 = _trunc(Decimal(10).ln(), 5)
        return _bc_str(_trunc(num / l10, 5))
    return "Inf"
