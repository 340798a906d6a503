// gather_chain.hip -- what does the memory system give a kernel made of per-lane DEPENDENT 32-byte gathers?
// That is the shape of a per-lane tree walk (rt_fastlanes.hpp: fetch a 32-byte node, test, pick the next node, fetch): every
// lane of every wave chases its own chain through a table of 32-byte records.  Measured here without the walk's arithmetic:
//   table of N records; a lane's next index comes from the record it has just read (so the fetches are dependent), taken
//   from a small set (the top of a tree: fits the L1) with probability `l1` per mille, from a hot set (fits the L2) with
//   probability `hot` per mille, else from the whole table;
//   W waves per SIMD (256-thread blocks, W per CU), `active` lanes of 64 taking part, `valu` dummy FMAs per step.
// Prints lane-steps per second for the chip.  usage: gather_chain [N_records] [hot_per_mille] [hot_records] [l1_per_mille] [l1_records]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ table, unsigned n, unsigned hot_n, unsigned hot_pm, unsigned l1_n, unsigned l1_pm,
                                             int steps, int active, int valu, unsigned *__restrict__ out)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned idx = (tid * 2654435761u) % n, acc = 0u;
    float f = (float)tid;
    if ((int)(threadIdx.x & 63) < active) {
        for (int s = 0; s < steps; ++s) {
            const uint4 a = table[2 * (size_t)idx], b = table[2 * (size_t)idx + 1];
            acc += a.y ^ b.z;
            for (int k = 0; k < valu; ++k) f = fmaf(f, 1.0000001f, 0.5f);
            unsigned r = a.x + (unsigned)s * 0x9E3779B9u + tid; // the record's content decides where the lane goes next
            r ^= r >> 16, r *= 0x7feb352du, r ^= r >> 15, r *= 0x846ca68bu, r ^= r >> 16;
            const unsigned dice = r % 1000u;
            idx = dice < l1_pm ? (r >> 10) % l1_n : dice < l1_pm + hot_pm ? (r >> 10) % hot_n : (r >> 10) % n;
        }
    }
    out[tid] = acc + (unsigned)f;
}

int main(int argc, char **argv)
{
    const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 1000000u;
    const unsigned hot_pm = argc > 2 ? (unsigned)atoi(argv[2]) : 850u;
    const unsigned hot_n = argc > 3 ? (unsigned)atoi(argv[3]) : 65536u;
    const unsigned l1_pm = argc > 4 ? (unsigned)atoi(argv[4]) : 0u;
    const unsigned l1_n = argc > 5 ? (unsigned)atoi(argv[5]) : 512u;
    std::vector<uint4> h(2 * (size_t)n);
    unsigned x = 12345u;
    for (auto &v : h) {
        x = x * 1664525u + 1013904223u; v.x = x;
        x = x * 1664525u + 1013904223u; v.y = x;
        x = x * 1664525u + 1013904223u; v.z = x;
        x = x * 1664525u + 1013904223u; v.w = x;
    }
    uint4 *d;
    unsigned *out;
    hipMalloc(&d, h.size() * sizeof(uint4));
    hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipMalloc(&out, sizeof(unsigned) * 256 * 8 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    printf("table %u records of 32 B (%.1f MB), %u per mille of the fetches go to the first %u records, %u per mille to the first %u\n", n, n * 32e-6,
           hot_pm, hot_n, l1_pm, l1_n);
    const int steps = 2048;
    for (int valu : {0, 60}) {
        for (int active : {64, 32}) {
            for (int w : {4, 6}) {
                const int blocks = 256 * w;
                chase<<<blocks, 256>>>(d, n, hot_n, hot_pm, l1_n, l1_pm, 64, active, valu, out);
                hipEventRecord(e0);
                chase<<<blocks, 256>>>(d, n, hot_n, hot_pm, l1_n, l1_pm, steps, active, valu, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double lane_steps = (double)blocks * 4 * active * steps;
                printf("valu/step %2d  lanes on %2d  waves/SIMD %d: %8.3f ms  %.3e lane-steps/s  (%.2f TB/s of 32-byte records)\n", valu, active, w, ms,
                       lane_steps / (ms * 1e-3), lane_steps * 32 / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
