// Thread-local error text behind fra_last_error().
#include <stdarg.h>

#include "common.h"

namespace fra {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_err; }

static std::atomic<long> g_alloc_calls{0};
hipError_t dev_malloc(void** p, size_t bytes) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc(p, bytes);
}
hipError_t dev_free(void* p) {
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    return hipFree(p);
}
long alloc_calls() { return g_alloc_calls.load(std::memory_order_relaxed); }

// ---- waits that give up: one host-visible word per device ---------------------------------------
static std::atomic<unsigned*> g_fault_host[64];
static std::atomic<unsigned*> g_fault_dev[64];
static std::atomic<int> g_spin_limit{-1};

unsigned* fault_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned* d = g_fault_dev[dev].load(std::memory_order_acquire);
    if (d) return d;
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    while (busy.test_and_set(std::memory_order_acquire)) {}
    d = g_fault_dev[dev].load(std::memory_order_acquire);
    if (!d) {
        unsigned* hp = nullptr;
        void* dp = nullptr;
        // pinned, mapped host memory: a kernel writes it directly, the host reads it without a copy or a synchronisation
        g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        if (hipHostMalloc(reinterpret_cast<void**>(&hp), 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hp) {
            *hp = 0;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
                g_fault_host[dev].store(hp, std::memory_order_release);
                d = static_cast<unsigned*>(dp);
                g_fault_dev[dev].store(d, std::memory_order_release);
            } else {
                (void)hipHostFree(hp);
            }
        }
        (void)hipGetLastError();
    }
    busy.clear(std::memory_order_release);
    return d;
}

int spin_limit(int dflt) {
    const int v = g_spin_limit.load(std::memory_order_relaxed);
    return v >= 0 ? v : dflt;
}

static unsigned take_fault(int device) {
    if (device < 0 || device >= 64) return 0;
    unsigned* hp = g_fault_host[device].load(std::memory_order_acquire);
    if (!hp) return 0;
    volatile unsigned* v = hp;
    const unsigned tag = *v;
    if (tag) *v = 0;
    return tag;
}

int check_device_fault(int device) {
    const unsigned tag = take_fault(device);
    if (!tag) return FRA_OK;
    set_error("a wait inside kernel '%c%c' (wave %u) gave up on device %d: the results of the launches before this call are invalid",
              (char)(tag >> 24), (char)((tag >> 16) & 0xff), tag & 0xffffu, device);
    return FRA_ERR_STATE;
}

static std::atomic<int> g_small_batch_mode{1};
int small_batch_mode() { return g_small_batch_mode.load(std::memory_order_relaxed); }
void set_small_batch_mode(int mode) { g_small_batch_mode.store(mode, std::memory_order_relaxed); }

}  // namespace fra

extern "C" {
const char* fra_last_error(void) { return fra::last_error(); }
int fra_version(void) { return 1; }
int fra_device_fault(int device, unsigned* tag_out) {
    const unsigned tag = fra::take_fault(device);
    if (tag_out) *tag_out = tag;
    return tag ? FRA_ERR_STATE : FRA_OK;
}
int fra_set_small_batch_mode(int mode) {
    if (mode != 0 && mode != 1) {
        fra::set_error("fra_set_small_batch_mode: mode %d (0 = batch-invariant bits, 1 = shortest small-batch launches)", mode);
        return FRA_ERR_ARG;
    }
    fra::set_small_batch_mode(mode);
    return FRA_OK;
}
int fra_get_small_batch_mode(void) { return fra::small_batch_mode(); }
long fra_dbg_alloc_calls(void) { return fra::alloc_calls(); }
int fra_dbg_set_spin_limit(int polls) {
    fra::g_spin_limit.store(polls, std::memory_order_relaxed);
    return FRA_OK;
}
}
