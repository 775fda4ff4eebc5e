// prot_internal.h -- entry points of libprot_hip.so that are NOT part of the drop-in boundary (include/prot_hip.h):
// hooks for the test-suite.  Exported so that a test process can call them; no product code does.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
// the shared library dlopen()ed in place of the system's RCCL by prot_allreduce_peaks (tests: a counting stand-in);
// only before the first all-reduce of the process, NULL or "" = the default search.  0 or PROT_EINVAL.
int prot_internal_set_rccl_library (const char* path);
#ifdef __cplusplus
}
#endif
