// gnx_io.h — internal interface between the HIP-free file code (gnx_io.cpp) and the C-ABI wrappers that need a context
// (gnx_api_vcf.hip).  Public contract: include/gnomix_io.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>

#include "../../include/gnomix_io.h"

// allocator of the genotype matrix (page-locked when a context is given); user = opaque
typedef void* (*gnx_io_alloc_fn)(void* user, size_t bytes);
typedef void (*gnx_io_free_fn)(void* user, void* p);

int gnx_io_vcf_read(const char* path, const char* region, int n_threads, gnx_io_alloc_fn alloc, gnx_io_free_fn release,
                    void* user, int pinned, gnx_vcf** out);
void gnx_io_set_error(const std::string& msg);

// process-wide worker pool: fn(tid) runs once on each of n workers (the caller is worker 0); returns when all are done
int gnx_io_threads(int requested);
void gnx_io_parallel(int n_workers, const std::function<void(int)>& fn);
