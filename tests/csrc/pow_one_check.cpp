// worker.cpp selectChildBySoftmaxCount skips std::pow when the exponent 1 / temperature is exactly 1 (the reference's default temperature):
// this checks that glibc's powf(x, 1.0f) returns x bit for bit — for every visit count a search can produce and for a sweep of other floats.
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>

int main()
{
    volatile float one = 1.0f; // keeps the compiler from folding the call
    long checked = 0;
    for (int c = 0; c <= 4000000; ++c) {
        const float x = static_cast<float>(c);
        const float y = std::pow(x, one);
        if (std::memcmp(&x, &y, 4) != 0) { printf("FAIL count %d\n", c); return 1; }
        ++checked;
    }
    for (uint32_t bits = 0x00800000u; bits < 0x7f800000u; bits += 9973u) { // positive normal floats
        float x; std::memcpy(&x, &bits, 4);
        const float y = std::pow(x, one);
        if (std::memcmp(&x, &y, 4) != 0) { printf("FAIL bits %08x\n", bits); return 1; }
        ++checked;
    }
    printf("OK %ld values\n", checked);
    return 0;
}
