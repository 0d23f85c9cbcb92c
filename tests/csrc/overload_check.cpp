// Which functions do the unqualified sqrt / fabs / powf of the reference's transformValue / invertValue (ref utils/utils.h:93-108) bind to?  The header includes
// <cmath> and no <math.h>: libstdc++ then declares the float overloads in namespace std only, and inside `namespace minizero::utils` the unqualified names find
// the C library's double functions.  This program is compiled with exactly utils.h's standard includes (its three Boost includes cannot be: Boost is absent)
// and prints the sizes of the results: "8 8" = double arithmetic, what oracle/o_nn.cpp, oracle/o_loader.cpp and the product spell out explicitly.
#include <algorithm>
#include <cmath>
#include <iomanip>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>
#include <cstdio>

namespace minizero::utils {
inline int sizes(float value) { return static_cast<int>(sizeof(sqrt(fabs(value) + 1))) * 10 + static_cast<int>(sizeof(fabs(value))); }
} // namespace minizero::utils

int main()
{
    const int s = minizero::utils::sizes(2.0f);
    printf("%d %d\n", s / 10, s % 10);
    return 0;
}
