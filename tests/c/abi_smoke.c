/* Plain-C consumer of the boundary: include/hrf.h must compile as C99 (-pedantic) and the library must be usable
 * through dlopen alone -- no C++, no torch types. Runs without a GPU: it only exercises the version query and the
 * argument validation that happens before any launch. Prints "ok" on success. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "hrf.h"

typedef int (*abi_fn)(void);
typedef const char* (*err_fn)(void);
typedef int (*scan_fn)(const void*, int, int64_t, int32_t*, int32_t*, hrf_stream_t);

int main(int argc, char** argv)
{
    void* h;
    abi_fn ver;
    err_fn err;
    scan_fn scan;
    if (argc < 2) return 2;
    h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    *(void**)(&ver) = dlsym(h, "hrf_abi_version");
    *(void**)(&err) = dlsym(h, "hrf_last_error");
    *(void**)(&scan) = dlsym(h, "hrf_scan_exclusive");
    if (!ver || !err || !scan) return 4;
    if (ver() != HRF_ABI_VERSION) return 5;
    if (scan(NULL, 0, -1, NULL, NULL, NULL) == 0) return 6;          /* rejected before any launch */
    if (strstr(err(), "hrf_scan_exclusive") == NULL) return 7;
    if (sizeof(hrf_level_meta) != 20 || sizeof(hrf_segment_meta) != 16 + 20 * HRF_MAX_LEVELS) return 8;
    if (sizeof(hrf_adam_tensor) != 56 || sizeof(hrf_grad_scaler) != 32) return 9;
    printf("ok\n");
    return 0;
}
