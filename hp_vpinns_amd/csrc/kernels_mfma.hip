// Placeholder until the MFMA kernels land: reports "not available" so the generic path runs.
#include "hpv_mfma.h"

struct HpvMfma { int dummy; };

HpvMfma* hpv_mfma_create(const NetDesc&, long, std::string* why) {
    if (why) *why = "MFMA path not built yet";
    return nullptr;
}
void hpv_mfma_destroy(HpvMfma* m) { delete m; }
int hpv_mfma_grad_rows(HpvMfma*) { return 0; }
void hpv_mfma_forward(HpvMfma*, const double*, const double*, double*, int, hipStream_t) {}
void hpv_mfma_backward(HpvMfma*, const double*, const double*, const double*, double*, int*, hipStream_t) {}
bool hpv_mfma_has_projection(HpvMfma*) { return false; }
void hpv_mfma_project(HpvMfma*, const ProjDesc&, const double*, double*, double*, const double*, const double*, long,
                      const double*, const double*, const double*, double*, double*, long, long, int, hipStream_t) {}
