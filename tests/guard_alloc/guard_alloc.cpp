// Guard-page allocator for torch.cuda.memory.CUDAPluggableAllocator (test infrastructure: tests/test_guard_pages_gpu.py).
// Every allocation gets its own virtual range [guard | mapped pages | guard] built with the HIP virtual-memory API; the guards stay UNMAPPED, and
// the tensor is placed flush against one of them (its end by default, its start with DSVT_GUARD_FRONT=1; 16-byte alignment), so that ANY access
// past that side of a buffer -- a read as much as a write -- is a GPU memory access fault that kills the process, instead of landing in a
// neighbour's block of the caching allocator.  (The write-side check of test_no_plugin_writes_outside_its_buffers cannot see reads.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
struct Block { hipDeviceptr_t base; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; };
std::map<void*, Block> g_blocks;
std::mutex g_mu;
size_t g_gran = 0;
FILE* g_log = nullptr;
bool g_front = false;
int g_fill = -1;          // DSVT_GUARD_FILL=<byte>: fresh allocations are filled with it (a buffer that is read before it is written shows up as a changed result)
unsigned long g_seq = 0;

void die(const char* what, hipError_t e) { fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e)); abort(); }
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    if (!g_gran) {
        hipError_t e = hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess) die("hipMemGetAllocationGranularity", e);
        g_front = getenv("DSVT_GUARD_FRONT") && atoi(getenv("DSVT_GUARD_FRONT")) != 0;
        if (const char* p = getenv("DSVT_GUARD_LOG")) g_log = fopen(p, "w");
        if (const char* p = getenv("DSVT_GUARD_FILL")) g_fill = atoi(p) & 0xff;
    }
    const size_t n = size > 0 ? (size_t)size : 1, mapped = (n + g_gran - 1) / g_gran * g_gran, reserved = mapped + 2 * g_gran;
    Block b{}; b.reserved = reserved; b.mapped = mapped;
    hipError_t e = hipMemAddressReserve(&b.base, reserved, g_gran, nullptr, 0);
    if (e != hipSuccess) die("hipMemAddressReserve", e);
    e = hipMemCreate(&b.h, mapped, &prop, 0);
    if (e != hipSuccess) die("hipMemCreate", e);
    char* lo = static_cast<char*>(b.base) + g_gran;
    e = hipMemMap(lo, mapped, 0, b.h, 0);
    if (e != hipSuccess) die("hipMemMap", e);
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(lo, mapped, &acc, 1);
    if (e != hipSuccess) die("hipMemSetAccess", e);
    if (g_fill >= 0) { (void)hipMemset(lo, g_fill, mapped); (void)hipDeviceSynchronize(); }
    char* p = g_front ? lo : lo + ((mapped - n) & ~(size_t)15);
    g_blocks[p] = b;
    if (g_log) { fprintf(g_log, "%lu %p %zu\n", g_seq, (void*)p, n); fflush(g_log); }
    ++g_seq;
    return p;
}

extern "C" void guard_free(void* ptr, ssize_t, int, hipStream_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_blocks.find(ptr);
    if (it == g_blocks.end()) return;
    static const int keep = getenv("DSVT_GUARD_KEEP") ? atoi(getenv("DSVT_GUARD_KEEP")) : 1;
    const Block b = it->second;
    g_blocks.erase(it);
    if (keep >= 2) return;                       // 2: nothing is ever released
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(static_cast<char*>(b.base) + g_gran, b.mapped);
    (void)hipMemRelease(b.h);
    // 1 (default): the virtual range is never handed out again -- a freed buffer stays an unmapped hole, so a use after free faults as well,
    // and no new mapping ever appears at an address the GPU has translated before (with reuse, DSVT_GUARD_KEEP=0, kernels were seen
    // reading the OLD pages of a re-mapped address on this stack: tools/dbg_tables.py)
    if (keep == 0) (void)hipMemAddressFree(b.base, b.reserved);
}
