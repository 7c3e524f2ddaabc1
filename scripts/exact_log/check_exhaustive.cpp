// check_exhaustive.cpp — ln_f32_cr (csrc/exact_log.h, host build) against x87 logl on EVERY positive finite
// binary32 input (subnormals included). Prints mismatch counts and writes the inputs whose logl value lies
// closest to a binary32 rounding boundary (where a 2^-63 error could still decide) for check_hard_cases.py.
//   g++ -O2 -std=c++17 -ffp-contract=off -fopenmp scripts/exact_log/check_exhaustive.cpp -o /tmp/xl_check && /tmp/xl_check hard.txt
#include <math.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "../../bevy_gaussian_splatting_amd/csrc/exact_log.h"

struct Hard { double dist; uint32_t bits; };

int main(int argc, char** argv) {
    const uint32_t first = 1u, last = 0x7F7FFFFFu;
    long long mism_logl = 0, mism_log = 0, mism_logf = 0;
    std::vector<Hard> hard;
#pragma omp parallel
    {
        long long a = 0, b = 0, c = 0;
        std::vector<Hard> mine;
#pragma omp for schedule(static, 1 << 16)
        for (long long i = first; i <= (long long)last; ++i) {
            uint32_t u = (uint32_t)i;
            float x; memcpy(&x, &u, 4);
            const float own = bgs::ln_f32_cr(x);
            const long double L = logl((long double)x);
            const float ref = (float)L;
            uint32_t ob, rb; memcpy(&ob, &own, 4); memcpy(&rb, &ref, 4);
            if (ob != rb) ++a;
            const float viaD = (float)log((double)x);
            uint32_t db; memcpy(&db, &viaD, 4);
            if (db != ob) ++b;
            const float viaF = logf(x);
            uint32_t fb; memcpy(&fb, &viaF, 4);
            if (fb != ob) ++c;
            if (L != 0.0L) {
                // distance of L to the nearer binary32 rounding boundary (midpoint of ref and its neighbour on L's side)
                const float nb = (long double)ref <= L ? nextafterf(ref, INFINITY) : nextafterf(ref, -INFINITY);
                const long double mid = ((long double)ref + (long double)nb) * 0.5L;
                const double dist = (double)(fabsl(L - mid) / fabsl(L));
                if (dist < 0x1p-55 || ob != rb || db != ob) mine.push_back(Hard{dist, u});  // incl. where binary64 log misrounds
            }
        }
#pragma omp critical
        { mism_logl += a; mism_log += b; mism_logf += c; hard.insert(hard.end(), mine.begin(), mine.end()); }
    }
    std::sort(hard.begin(), hard.end(), [](const Hard& p, const Hard& q) { return p.dist < q.dist; });
    printf("inputs %lld\n", (long long)last - first + 1);
    printf("ln_f32_cr != (float)logl(x)        : %lld\n", mism_logl);
    printf("ln_f32_cr != (float)log((double)x) : %lld\n", mism_log);
    printf("ln_f32_cr != logf(x)               : %lld\n", mism_logf);
    printf("inputs within 2^-55 (relative) of a rounding boundary per logl: %zu; closest 2^%.2f\n", hard.size(),
           hard.empty() ? 0.0 : log2(hard[0].dist));
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        for (size_t k = 0; k < hard.size() && k < 4096; ++k) fprintf(f, "%08x %.3e\n", hard[k].bits, hard[k].dist);
        fclose(f);
    }
    return mism_logl != 0;
}
