// Checks the division shortcuts of the PUCT selection kernel (pool_body.h) against real IEEE division on the host:
//   f32:  a / c            == (float)((double)a * RN64(1 / c))                       for integer-valued c in [1, 4096]
//   f64:  x / d            == fma(fma(-d, q0, x), r, q0), q0 = x * r, r = RN64(1/d)   for integer-valued d in [1, 4096]  (Markstein)
// IEEE mul / fma / conversions are deterministic, so the host result is the device result.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

int main()
{
    std::mt19937_64 gen(99);
    long bad32 = 0, bad64 = 0, n32 = 0, n64 = 0;
    for (int c = 1; c <= 4096; ++c) {
        const double r = 1.0 / static_cast<double>(c);
        const float cf = static_cast<float>(c);
        const double cd = static_cast<double>(c);
        for (int it = 0; it < 6000; ++it) {
            // f32 numerators the kernel sees: value * count with value in [-1, 1] (any float), plus raw random bit patterns
            uint32_t bits = static_cast<uint32_t>(gen());
            float a;
            if (it % 3 == 0) {
                bits = (bits & 0x807FFFFFu) | ((100u + bits % 40u) << 23); // exponents 2^-27 .. 2^12
                memcpy(&a, &bits, 4);
            } else {
                const float v = static_cast<float>((static_cast<double>(gen() >> 11) / 9007199254740992.0) * 2.0 - 1.0);
                a = v * cf;
            }
            const float want = a / cf;
            const float got = static_cast<float>(static_cast<double>(a) * r);
            ++n32;
            if (memcmp(&want, &got, 4) != 0) { if (bad32 < 5) { printf("f32 mismatch a=%a c=%d want=%a got=%a\n", a, c, want, got); } ++bad32; }
            // f64 numerators: (double)(bias * policy) * sqrt(N)
            const float bp = static_cast<float>(static_cast<double>(gen() >> 11) / 9007199254740992.0 * 3.0);
            const double x = static_cast<double>(bp) * std::sqrt(static_cast<double>(1 + gen() % 2000));
            const double wantd = x / cd;
            const double q0 = x * r;
            const double gotd = std::fma(std::fma(-cd, q0, x), r, q0);
            ++n64;
            if (memcmp(&wantd, &gotd, 8) != 0) { if (bad64 < 5) { printf("f64 mismatch x=%a d=%d want=%a got=%a\n", x, c, wantd, gotd); } ++bad64; }
        }
    }
    printf("%s f32 %ld/%ld bad, f64 %ld/%ld bad\n", (bad32 || bad64) ? "FAIL" : "OK", bad32, n32, bad64, n64);
    return (bad32 || bad64) ? 1 : 0;
}
