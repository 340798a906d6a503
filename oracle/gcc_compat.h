/* Force-included (-include) when the reference sources are compiled with g++.
 * The reference was written against MSVC, whose <cmath> puts the float libm
 * entry points in namespace std.  libstdc++ 11 does not, so three call sites
 * (std::sinf, std::cosf, std::logf) need these using-declarations.  Nothing
 * here changes arithmetic: the names resolve to the same glibc functions. */
#pragma once
#include <math.h>
namespace std { using ::sinf; using ::cosf; using ::logf; }
