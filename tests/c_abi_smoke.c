/* Plain-C consumer of include/sdrk.h: proves the header is valid C99 and that the shared library can be
 * bound without C++ or Python.  Built and run by tests/test_abi.py (CPU: no compute calls); on a GPU box it
 * also transforms one frame.  Usage: c_abi_smoke [gpu] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrk.h"

#define PI 3.14159265358979323846

int main(int argc, char** argv) {
    if (sdrk_version() != SDRK_VERSION) { fprintf(stderr, "version mismatch\n"); return 1; }
    if (sdrk_last_error() == NULL) { fprintf(stderr, "last_error is NULL\n"); return 1; }
    int n = sdrk_device_count();
    printf("sdrk %d, %d device(s)\n", sdrk_version(), n);
    sdrk_plan* plan = NULL;
    int st = sdrk_plan_create(0, 4096, 16, SDRK_WINDOW_RECT, NULL, 1e-12f, 1, &plan);
    if (n == 0) {
        if (st != SDRK_ERR_NO_DEVICE || plan != NULL) { fprintf(stderr, "expected NO_DEVICE, got %d\n", st); return 1; }
        printf("no device: %s\n", sdrk_last_error());
        return 0;
    }
    if (st != SDRK_OK) { fprintf(stderr, "plan_create: %s\n", sdrk_last_error()); return 1; }
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        float* iq = (float*)calloc(2 * 4096, sizeof(float));
        float* db = (float*)malloc(4096 * sizeof(float));
        for (int i = 0; i < 4096; ++i) {                       /* on-bin tone at k = +100 */
            iq[2 * i] = (float)cos(2.0 * PI * 100.0 * i / 4096.0);
            iq[2 * i + 1] = (float)sin(2.0 * PI * 100.0 * i / 4096.0);
        }
        st = sdrk_exec_host(plan, iq, 1, 4096, db);
        if (st != SDRK_OK) { fprintf(stderr, "exec_host: %s\n", sdrk_last_error()); return 1; }
        int arg = 0;
        for (int i = 1; i < 4096; ++i) if (db[i] > db[arg]) arg = i;
        printf("peak %.4f dB at index %d\n", db[arg], arg);
        if (arg != 2048 + 100 || fabs(db[arg] - 20.0 * log10(4096.0)) > 1e-3) return 1;
        free(iq); free(db);
    }
    return sdrk_plan_destroy(plan) == SDRK_OK ? 0 : 1;
}
