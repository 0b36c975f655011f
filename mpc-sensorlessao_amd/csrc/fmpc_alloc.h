// Counted device allocation (fmpc_alloc_generation, include/fastmpc.h).
// Every device allocation and release of the library's host files (fmpc_api.hip, fmpc_est_api.hip) bumps one process-wide
// counter: a HIP graph recorded over solve / estimator calls holds the addresses of the handles' workspaces, and a replay after
// any of them has been reallocated would use freed memory -- whoever replays compares the counter first (RecordedSolves in
// recorded.py).  Include this header AFTER <hip/hip_runtime.h>: it redirects hipMalloc / hipFree of the including file.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

extern std::atomic<unsigned long long> fmpc_alloc_gen;       // defined in fmpc_api.hip
static inline hipError_t fmpc_counted_malloc(void** p, size_t bytes) { fmpc_alloc_gen.fetch_add(1); return hipMalloc(p, bytes); }
static inline hipError_t fmpc_counted_free(void* p) { fmpc_alloc_gen.fetch_add(1); return hipFree(p); }
#define hipMalloc(p, bytes) fmpc_counted_malloc((void**)(p), (bytes))
#define hipFree(p) fmpc_counted_free((void*)(p))
