// gnx_io.h — internal interface of the HIP-free file code: gnx_io.cpp (worker pool, number text, writers), gnx_vcf.cpp (VCF
// reader) and the C-ABI wrappers that need a context (gnx_api_vcf.hip).  Public contract: include/gnomix_io.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/gnomix_io.h"

// allocator of the genotype matrix (page-locked when a context is given); user = opaque
typedef void* (*gnx_io_alloc_fn)(void* user, size_t bytes);
typedef void (*gnx_io_free_fn)(void* user, void* p);

int gnx_io_vcf_read(const char* path, const char* region, int n_threads, gnx_io_alloc_fn alloc, gnx_io_free_fn release,
                    void* user, int pinned, gnx_vcf** out);
int gnx_io_fail(int code, const std::string& msg);  // sets the thread's message, returns code
int gnx_io_write_file(const char* path, const char* head, size_t head_len, const char* body, size_t body_len);  // create / truncate, head then body

// process-wide worker pool: fn(tid) runs once on each of n workers (the caller is worker 0); returns when all are done
int gnx_io_threads(int requested);         // <= 0: every core this process may run on (GNX_IO_THREADS overrides)
int gnx_io_stream_threads(int requested);  // <= 0: the same, capped where memory streaming stops scaling
int gnx_io_cpu_quota();                    // CPUs the container's cgroup allows (0: unlimited)
void gnx_io_parallel(int n_workers, const std::function<void(int)>& fn);
double gnx_io_now();

// the parsed file (gnx_vcf.cpp owns it; the phased-VCF writer in gnx_io.cpp reads its columns)
struct gnx_strcol {
  std::string blob;
  std::vector<int64_t> off{0};
};
struct gnx_vcf_ovf {  // an allele >= 2: code 3 in the 2-bit matrix, the number here
  int64_t row;
  int32_t hap, allele;
};
struct gnx_vcf {
  gnx_vcf_info info{};
  std::vector<int64_t> pos;
  std::vector<float> qual;
  gnx_strcol col[8];
  uint8_t* gt2 = nullptr;
  gnx_io_free_fn release = nullptr;
  void* user = nullptr;
  std::vector<gnx_vcf_ovf> ovf;
};
