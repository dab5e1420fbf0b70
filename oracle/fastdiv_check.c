/* oracle/fastdiv_check.c -- TEST INFRASTRUCTURE ONLY.
 * Evidence for the division used on the critical path of k_centers (pbnet_amd/csrc/cluster.hip): with y = RN(1/n),
 *     q0 = RN(d*y);  r = fma(-q0, n, d);  q1 = fma(r, y, q0)
 * equals the IEEE quotient RN(d/n) for every integer n whose 24-bit significand is not all ones (Markstein's
 * correction-step theorem).  This program compares the two bit for bit on random and near-halfway dividends.
 * usage: fastdiv_check [max_n] [trials_per_n]     (defaults 400000, 600: 640 M comparisons, 0 mismatches measured) */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static inline float fastdiv(float d, float n, float y) { float q = d * y; float r = fmaf(-q, n, d); return fmaf(r, y, q); }
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv) {
    long long bad = 0, total = 0;
    const int max_n = argc > 1 ? atoi(argv[1]) : 400000, trials = argc > 2 ? atoi(argv[2]) : 600;
    for (int N = 1; N <= max_n; ++N) {
        float n = (float)N, y = 1.0f / n;
        for (int t = 0; t < trials; ++t) {
            uint32_t bits = (uint32_t)rnd();
            int e = 90 + (int)(rnd() % 60);            /* exponents 2^-37 .. 2^22 */
            bits = (bits & 0x807fffffu) | ((uint32_t)e << 23);
            float d; memcpy(&d, &bits, 4);
            float want = d / n, got = fastdiv(d, n, y);
            total++;
            if (memcmp(&want, &got, 4)) { if (bad < 5) printf("mismatch N=%d d=%a want=%a got=%a\n", N, d, want, got); bad++; }
        }
        /* adversarial: d close to n*q for q with many trailing ones / halfway cases */
        for (int t = 0; t < trials / 3; ++t) {
            uint32_t qb = ((uint32_t)rnd() & 0x007fffffu) | ((uint32_t)(100 + rnd() % 40) << 23);
            float q; memcpy(&q, &qb, 4);
            double prod = (double)q * (double)n;        /* exact in double */
            float d0 = (float)prod;
            for (int k = -2; k <= 2; ++k) {
                float d = d0;
                for (int j = 0; j < (k < 0 ? -k : k); ++j) d = nextafterf(d, k < 0 ? -INFINITY : INFINITY);
                float want = d / n, got = fastdiv(d, n, y);
                total++;
                if (memcmp(&want, &got, 4)) { if (bad < 5) printf("mismatch(adv) N=%d d=%a want=%a got=%a\n", N, d, want, got); bad++; }
            }
        }
    }
    printf("total %lld mismatches %lld\n", total, bad);
    return bad != 0;
}
