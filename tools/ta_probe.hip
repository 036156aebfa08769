// ta_probe.hip -- what a divergent 16-B load costs on gfx950's vector memory path, by access
// shape.  Every wave issues `iters` rounds of loads from a table at pseudo-random record
// indices (one record per lane, per lane pair or per lane quad) and sums what it reads.
// Printed per variant: ns per wave-instruction-equivalent and lane-loads per CU-clock.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ta_probe.hip -o build_ab/ta_probe && build_ab/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s)
{
    s = s * 747796405u + 2891336453u;
    uint32_t v = s;
    v ^= v >> ((v >> 28u) + 4u);
    v *= 277803737u;
    return v ^ (v >> 22u);
}

// MODE 0: one 16-B load per lane, every lane its own 32-B record (first half)
// MODE 1: two 16-B loads per lane, both halves of its own 32-B record
// MODE 2: one 16-B load per lane, lane pairs share a record (even lane half 0, odd lane half 1)
// MODE 3: two loads per lane the pair way: first the even lane's record, then the odd lane's
// MODE 4: one 16-B load per lane, quads share a 64-B record
// MODE 5: one 8-B load per lane, own record
// MODE 6: one 4-B load per lane, own record
// MODE 7: one 16-B LDS read per lane from a 32-KB table at random records
// MODE 8: two 16-B LDS reads per lane (both halves of a 32-B record)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ table, uint32_t mask, int iters, uint32_t* out)
{
    __shared__ uint4 lds[2048]; // 32 KB
    if (MODE >= 7) {
        for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x)
            lds[i] = table[i];
        __syncthreads();
    }
    uint32_t s = blockIdx.x * 256u + threadIdx.x + 12345u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r = lcg(s) & mask; // record index (32-B records)
        if (MODE == 0) {
            const uint4 a = table[2u * r];
            acc += a.x ^ a.w;
        } else if (MODE == 1) {
            const uint4 a = table[2u * r], b = table[2u * r + 1u];
            acc += a.x ^ a.w ^ b.y;
        } else if (MODE == 2) {
            const uint32_t rr = __shfl(r, lane & ~1u);
            const uint4 a = table[2u * rr + (lane & 1u)];
            acc += a.x ^ a.w;
        } else if (MODE == 3) {
            const uint32_t r0 = __shfl(r, lane & ~1u), r1 = __shfl(r, lane | 1u);
            const uint4 a = table[2u * r0 + (lane & 1u)], b = table[2u * r1 + (lane & 1u)];
            acc += a.x ^ a.w ^ b.y;
        } else if (MODE == 4) {
            const uint32_t rr = __shfl(r, lane & ~3u) & ~1u;
            const uint4 a = table[2u * rr + (lane & 3u)];
            acc += a.x ^ a.w;
        } else if (MODE == 5) {
            const uint2 a = *reinterpret_cast<const uint2*>(table + 2u * r);
            acc += a.x ^ a.y;
        } else if (MODE == 6) {
            acc += *reinterpret_cast<const uint32_t*>(table + 2u * r);
        } else if (MODE == 7) {
            const uint4 a = lds[2u * (r & 1023u)];
            acc += a.x ^ a.w;
        } else if (MODE == 8) {
            const uint4 a = lds[2u * (r & 1023u)], b = lds[2u * (r & 1023u) + 1u];
            acc += a.x ^ a.w ^ b.y;
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <int MODE>
static void run(const char* name, const uint4* table, uint32_t records, int loadsPerLane, uint32_t* out, int numCU)
{
    const int iters = 2000, blocks = numCU * 6;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, table, records - 1u, 50, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, table, records - 1u, iters, out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double laneLoads = (double)blocks * 256 * iters * loadsPerLane;
    const double perCuClk = laneLoads / numCU / (ms * 1e-3 * 2.4e9);
    printf("%-58s table %8.2f MB  %8.3f ms  %6.3f lane-loads per CU-clock (2.4 GHz)  %7.1f Glane-loads/s\n", name, records * 32.0 / 1e6, ms,
        perCuClk, laneLoads / (ms * 1e-3) / 1e9);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int numCU = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, numCU, p.clockRate);
    uint32_t* out;
    CHECK(hipMalloc(&out, 4));
    for (uint32_t records : { 512u, 65536u, 1u << 20 }) { // 16 KB (L1), 2 MB (L2), 32 MB (Infinity Cache)
        std::vector<uint4> h(2 * (size_t)records);
        for (size_t i = 0; i < h.size(); i++)
            h[i] = make_uint4((uint32_t)i * 2654435761u, 1, 2, (uint32_t)i);
        uint4* table;
        CHECK(hipMalloc(&table, h.size() * 16));
        CHECK(hipMemcpy(table, h.data(), h.size() * 16, hipMemcpyHostToDevice));
        run<0>("0: 16 B per lane, own record", table, records, 1, out, numCU);
        run<1>("1: 2 x 16 B per lane, own 32-B record", table, records, 2, out, numCU);
        run<2>("2: 16 B per lane, lane pairs share a 32-B record", table, records, 1, out, numCU);
        run<3>("3: 2 x 16 B per lane, pair-wise (two records per pair)", table, records, 2, out, numCU);
        run<4>("4: 16 B per lane, quads share a 64-B record", table, records, 1, out, numCU);
        run<5>("5: 8 B per lane, own record", table, records, 1, out, numCU);
        run<6>("6: 4 B per lane, own record", table, records, 1, out, numCU);
        if (records == 65536u) {
            run<7>("7: 16 B per lane from LDS, random of 1024 records", table, 2048, 1, out, numCU);
            run<8>("8: 2 x 16 B per lane from LDS, random of 1024 records", table, 2048, 2, out, numCU);
        }
        CHECK(hipFree(table));
    }
    return 0;
}
