"""Generate the polynomial coefficients of the deterministic fp64 math kernels (det_math.h) with mpmath.

The RANSAC path is compared bit-for-bit between the CPU oracle (gcc) and the HIP kernels, so it may not call
libm / ocml transcendental functions (they differ in the last bits).  sin, cos, asin, exp are evaluated with the
polynomials printed here using only + - * (IEEE-exact on both sides, contraction disabled).
Usage: python tools/gen_det_math.py   -> prints C arrays with hex-float literals.
"""
import mpmath as mp

mp.mp.dps = 60


def cheb_poly(f, a, b, n):
    # mpmath.chebyfit returns coefficients highest degree first
    c, err = mp.chebyfit(f, [a, b], n, error=True)
    return c[::-1], err


def emit(name, coeffs, err, comment):
    print(f"// {comment}; max fit error {mp.nstr(err, 3)}")
    print(f"ACEZ_DET_CONST double {name}[{len(coeffs)}] = {{")
    for c in coeffs:
        print(f"    {float(c).hex()},  // {mp.nstr(c, 20)}")
    print("};")


q = (mp.pi / 4) ** 2
# sin(x) = x * S(x^2), cos(x) = C(x^2) on |x| <= pi/4 (slightly widened)
S, es = cheb_poly(lambda t: mp.sin(mp.sqrt(t)) / mp.sqrt(t) if t != 0 else mp.mpf(1), 0, q * 1.05, 10)
Cc, ec = cheb_poly(lambda t: mp.cos(mp.sqrt(t)), 0, q * 1.05, 10)
# asin(z) = z * A(z^2) on |z| <= 0.5 (widened to 0.52)
A, ea = cheb_poly(lambda t: mp.asin(mp.sqrt(t)) / mp.sqrt(t) if t != 0 else mp.mpf(1), 0, 0.52 ** 2, 20)
# exp(r) on |r| <= ln2/2
E, ee = cheb_poly(lambda r: mp.exp(r), -mp.log(2) / 2 * 1.02, mp.log(2) / 2 * 1.02, 14)
emit("DET_SIN_C", S, es, "sin(x)/x as a polynomial in x^2, |x| <= pi/4")
emit("DET_COS_C", Cc, ec, "cos(x) as a polynomial in x^2, |x| <= pi/4")
emit("DET_ASIN_C", A, ea, "asin(z)/z as a polynomial in z^2, |z| <= 0.5")
emit("DET_EXP_C", E, ee, "exp(r), |r| <= ln2/2")
pio2 = mp.pi / 2
hi = mp.mpf(float(pio2))
# split pi/2 so that k*hi is exact for |k| < 2^20: keep 33 significant bits
import math
hi33 = math.ldexp(round(math.ldexp(float(pio2), 32)), -32)
lo = pio2 - mp.mpf(hi33)
print("// pi/2 = PIO2_HI + PIO2_LO, PIO2_HI has 33 significant bits")
print("PIO2_HI", float(hi33).hex(), "PIO2_LO", float(lo).hex())
ln2 = mp.log(2)
l33 = math.ldexp(round(math.ldexp(float(ln2), 33)), -33)
print("LN2_HI", float(l33).hex(), "LN2_LO", float(ln2 - mp.mpf(l33)).hex())
print("INV_LN2", float(1 / ln2).hex(), "TWO_OVER_PI", float(2 / mp.pi).hex(), "PI", float(mp.pi).hex(), "PIO2", float(pio2).hex())
