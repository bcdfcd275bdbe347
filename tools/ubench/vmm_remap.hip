// Kernel-free reproducer (HIP runtime calls only): is a virtual address that was mapped, unmapped and mapped again to OTHER physical pages
// translated afresh?  Round 4 saw kernels and copies use the old pages of a re-mapped address (tests/guard_alloc with DSVT_GUARD_KEEP=0,
// tools/dbg_tables.py): each iteration maps fresh pages at a (usually recycled) address, fills them with a new byte, reads them back.
//   hipcc -O2 tools/ubench/vmm_remap.hip -o /tmp/vmm_remap && /tmp/vmm_remap [iterations] [buffers in flight]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, live = argc > 2 ? atoi(argv[2]) : 4;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    struct B { hipDeviceptr_t p; hipMemGenericAllocationHandle_t h; unsigned char v; };
    std::vector<B> ring;
    std::vector<unsigned char> host(gran);
    int bad = 0, reused = 0; std::vector<void*> seen;
    for (int i = 0; i < iters; ++i) {
        B b{}; b.v = (unsigned char)(1 + i % 250);
        CK(hipMemAddressReserve(&b.p, gran, gran, nullptr, 0)); CK(hipMemCreate(&b.h, gran, &prop, 0)); CK(hipMemMap(b.p, gran, 0, b.h, 0)); CK(hipMemSetAccess(b.p, gran, &acc, 1));
        for (void* q : seen) if (q == b.p) { ++reused; break; }
        seen.push_back(b.p);
        CK(hipMemset(b.p, b.v, gran)); CK(hipDeviceSynchronize());
        ring.push_back(b);
        for (const B& r : ring) {                       // every live buffer must still hold its own byte
            CK(hipMemcpy(host.data(), r.p, gran, hipMemcpyDeviceToHost));
            size_t wrong = 0; for (unsigned char c : host) wrong += c != r.v;
            if (wrong) { if (bad < 8) printf("iteration %d: buffer at %p holds %zu bytes that are not 0x%02x (first byte 0x%02x)\n", i, r.p, wrong, r.v, host[0]); ++bad; }
        }
        if ((int)ring.size() > live) { B o = ring.front(); ring.erase(ring.begin()); CK(hipMemUnmap(o.p, gran)); CK(hipMemRelease(o.h)); CK(hipMemAddressFree(o.p, gran)); }
    }
    printf("granularity %zu, %d iterations, %d re-used addresses, %d wrong read-backs\n", gran, iters, reused, bad);
    return 0;
}
