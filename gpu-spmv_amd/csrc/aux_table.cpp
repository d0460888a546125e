// aux_table.cpp — side tables for per-matrix auxiliary device data.
//
// CSRMatrix / ELLMatrix keep the reference's layout, so what the kernels
// precompute (row statistics, merge-path tile table, ELL slot count) is keyed
// by the matrix's device index array and dropped in csr_free_gpu / ell_free_gpu.
#include "internal.h"
#include "tiled.h"

#include <dlfcn.h>

#include <cstdlib>

#include <cstring>
#include <algorithm>
#include <mutex>
#include <unordered_map>

namespace spmv {
namespace detail {

namespace {

std::mutex g_lock;
std::unordered_map<const void*, CsrAux*>& csr_table() {
    static auto* t = new std::unordered_map<const void*, CsrAux*>();
    return *t;
}
std::unordered_map<const void*, EllAux*>& ell_table() {
    static auto* t = new std::unordered_map<const void*, EllAux*>();
    return *t;
}

void release(CsrAux* a) {
    a->pagerank.release();
    if (a->d_tile_rows) (void)hipFree(a->d_tile_rows);
    if (a->d_carry_row) (void)hipFree(a->d_carry_row);
    if (a->d_carry_val) (void)hipFree(a->d_carry_val);
    for (const CsrAux::MergeCarry& c : a->extra_carry) {
        if (c.row) (void)hipFree(c.row);
        if (c.val) (void)hipFree(c.val);
    }
    a->tiled.reset();        // (users still holding the plan keep it alive)
    delete a;
}

} // namespace

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx& roctx() {
    static Roctx api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the marker library a profiler already brought into the process, or — on request (SPMV_ROCTX=1, e.g.
        // under `rocprofv3 --marker-trace`) — loaded here; never loaded into an unprofiled process
        void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_NOLOAD);
        if (!lib && std::getenv("SPMV_ROCTX")) {
            lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        }
        if (!lib) return;
        api.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        api.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!api.push || !api.pop) api = Roctx();
    });
    return api;
}
} // namespace

TraceRange::TraceRange(const char* name) : open_(false) {
    const Roctx& api = roctx();
    if (api.push) {
        (void)api.push(name);
        open_ = true;
    }
}

TraceRange::~TraceRange() {
    if (open_) (void)roctx().pop();
}

hipError_t malloc_any_time(void** ptr, size_t bytes) {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool switched = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    const hipError_t e = hipMalloc(ptr, bytes);
    if (switched) (void)hipThreadExchangeStreamCaptureMode(&mode);
    return e;
}

void PrWorkspace::release() {
    for (void* p : {static_cast<void*>(r[0]), static_cast<void*>(r[1]), static_cast<void*>(mask), static_cast<void*>(partials),
                    static_cast<void*>(sums), state, static_cast<void*>(dangling_count)}) {
        if (p) (void)hipFree(p);
    }
    if (pinned_state) (void)hipHostFree(pinned_state);
    if (pinned_ranks) (void)hipHostFree(pinned_ranks);
    for (hipEvent_t e : seen) if (e) (void)hipEventDestroy(e);
    *this = PrWorkspace();
}

bool debug_option(const char* key, long long* value, char* text, size_t text_size) {
    return list_option(std::getenv("SPMV_DEBUG"), key, value, text, text_size);
}

bool list_option(const char* env, const char* key, long long* value, char* text, size_t text_size) {
    if (!env || !*env) return false;
    const size_t klen = std::strlen(key);
    for (const char* p = env; *p;) {
        const char* end = std::strchr(p, ',');
        const size_t len = end ? static_cast<size_t>(end - p) : std::strlen(p);
        if (len >= klen && std::strncmp(p, key, klen) == 0 && (len == klen || p[klen] == '=')) {
            const char* v = len > klen ? p + klen + 1 : "";
            const size_t vlen = len > klen ? len - klen - 1 : 0;
            if (value) *value = vlen ? std::atoll(v) : 1;
            if (text && text_size) {
                const size_t n = std::min(vlen, text_size - 1);
                std::memcpy(text, v, n);
                text[n] = '\0';
            }
            return true;
        }
        if (!end) break;
        p = end + 1;
    }
    return false;
}

bool debug_is(const char* key, const char* expected) {
    char text[32];
    return debug_option(key, nullptr, text, sizeof(text)) && std::strcmp(text, expected) == 0;
}

CsrAux* aux_lookup(const void* key, bool create) {
    if (!key) return nullptr;
    std::lock_guard<std::mutex> guard(g_lock);
    auto& t = csr_table();
    auto it = t.find(key);
    if (it != t.end()) return it->second;
    if (!create) return nullptr;
    CsrAux* a = new CsrAux();
    t.emplace(key, a);
    return a;
}

void aux_drop(const void* key) {
    if (!key) return;
    CsrAux* victim = nullptr;
    {
        std::lock_guard<std::mutex> guard(g_lock);
        auto& t = csr_table();
        auto it = t.find(key);
        if (it == t.end()) return;
        victim = it->second;
        t.erase(it);
    }
    release(victim);
}

namespace {
std::mutex g_build_lock;   // one plan build at a time (two threads may meet on the same matrix)
}

namespace {
bool plan_matches(const std::shared_ptr<TiledPlan>& p, const CSRMatrix* A) {
    return p->num_rows == A->num_rows && p->num_cols == A->num_cols && p->csr_nnz == A->nnz &&
           p->csr_cols == A->d_col_indices && p->csr_vals == A->d_values;
}
std::shared_ptr<TiledPlan> adopt(TiledPlan* raw) {
    return std::shared_ptr<TiledPlan>(raw, [](TiledPlan* p) { tiled_free(p); });
}
}

PlanRef tiled_plan_if_cached(const CSRMatrix* A) {
    if (!A || !A->d_row_ptrs) return nullptr;
    std::lock_guard<std::mutex> building(g_build_lock);
    CsrAux* aux = aux_lookup(A->d_row_ptrs, false);
    if (aux && aux->tiled && plan_matches(aux->tiled, A)) return aux->tiled;
    return nullptr;
}

PlanRef tiled_plan_for(const CSRMatrix* A, hipStream_t s) {
    if (!A || !A->d_row_ptrs || !tiled_eligible(A)) return nullptr;
    std::lock_guard<std::mutex> building(g_build_lock);
    CsrAux* aux = aux_lookup(A->d_row_ptrs, true);
    if (aux->tiled && !plan_matches(aux->tiled, A)) {
        aux->tiled.reset();           // header or arrays changed under the same row-pointer array
        aux->tiled_failed = false;
        ++aux->plan_replacements;     // (two matrices taking turns over one row-pointer array: promotion gives up on them)
    }
    if (!aux->tiled && !aux->tiled_failed) {
        const TraceRange range("spmv:tiled_plan_build");
        TiledPlan* built = nullptr;
        if (tiled_build(A, &built, s) != hipSuccess) {
            (void)hipGetLastError();
            aux->tiled_failed = true;
        } else {
            aux->tiled = adopt(built);
        }
    }
    return aux->tiled;
}

EllAux* ell_aux_lookup(const void* key, bool create) {
    if (!key) return nullptr;
    std::lock_guard<std::mutex> guard(g_lock);
    auto& t = ell_table();
    auto it = t.find(key);
    if (it != t.end()) return it->second;
    if (!create) return nullptr;
    EllAux* a = new EllAux();
    t.emplace(key, a);
    return a;
}

void ell_aux_drop(const void* key) {
    if (!key) return;
    std::lock_guard<std::mutex> guard(g_lock);
    auto& t = ell_table();
    auto it = t.find(key);
    if (it == t.end()) return;
    delete it->second;
    t.erase(it);
}

PlanRef tiled_plan_for(const ELLMatrix* A, hipStream_t s) {
    if (!A || !A->d_col_indices || !A->d_values || !tiled_eligible(A)) return nullptr;
    std::lock_guard<std::mutex> building(g_build_lock);
    EllAux* aux = ell_aux_lookup(A->d_col_indices, true);
    if (aux->tiled && (aux->tiled->num_rows != A->num_rows || aux->tiled->num_cols != A->num_cols ||
                       aux->tiled->csr_nnz != static_cast<long long>(A->num_rows) * A->max_nnz_per_row ||
                       aux->tiled->csr_vals != A->d_values)) {
        aux->tiled.reset();
        aux->tiled_failed = false;
    }
    if (!aux->tiled && !aux->tiled_failed) {
        TiledPlan* built = nullptr;
        if (tiled_build(A, &built, s) != hipSuccess) {
            (void)hipGetLastError();
            aux->tiled_failed = true;
        } else {
            aux->tiled = adopt(built);
        }
    }
    return aux->tiled;
}

} // namespace detail
} // namespace spmv
