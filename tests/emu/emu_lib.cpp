// tests/emu/emu_lib.cpp — TEST INFRASTRUCTURE ONLY: the product's kernels + pipeline compiled for the
// host wave emulator (hip_emu.hpp) and exported with the same call shapes as the C-ABI device entry points.
#include "hip_emu.hpp"

#include "../../crystals-kyber_amd/csrc/mlkem_pipeline.hpp"

#include <stdlib.h>

using namespace mlkem;

static Workspace make_ws(size_t n) {
    Workspace ws;
    ws.A = (uint16_t*)aligned_alloc(64, (n * 16 * 512 + 63) / 64 * 64 + 64);
    ws.prf = (uint8_t*)aligned_alloc(64, n * 9 * 192 + 64);
    ws.r = (uint8_t*)aligned_alloc(64, n * 32 + 64);
    ws.rho = (uint8_t*)aligned_alloc(64, n * 32 + 64);
    ws.m = (uint8_t*)aligned_alloc(64, n * 32 + 64);
    ws.Kp = (uint8_t*)aligned_alloc(64, n * 32 + 64);
    ws.Kbar = (uint8_t*)aligned_alloc(64, n * 32 + 64);
    ws.cap_items = n;
    return ws;
}
static void free_ws(Workspace& ws) {
    free(ws.A); free(ws.prf); free(ws.r); free(ws.rho); free(ws.m); free(ws.Kp); free(ws.Kbar);
}

extern "C" {
int emu_keygen(int set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk) {
    Workspace ws = make_ws(n);
    int rc = keygen_dispatch(nullptr, set, n, d, z, ek, dk, ws);
    free_ws(ws);
    return rc;
}
int emu_encaps(int set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K) {
    Workspace ws = make_ws(n);
    int rc = encaps_dispatch(nullptr, set, n, ek, m, c, K, ws);
    free_ws(ws);
    return rc;
}
int emu_decaps(int set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status, int hash_check) {
    Workspace ws = make_ws(n);
    int rc = decaps_dispatch(nullptr, set, n, dk, c, K, status, hash_check != 0, ws);
    free_ws(ws);
    return rc;
}
void emu_ntt(int inverse, size_t n, const uint16_t* in, uint16_t* out) { ntt_launch(nullptr, inverse != 0, n, in, out); }
void emu_basemul(size_t n, const uint16_t* a, const uint16_t* b, uint16_t* h) { basemul_launch(nullptr, n, a, b, h); }
int emu_cbd(int eta, size_t n, const uint8_t* bytes, uint16_t* out) { return cbd_launch(nullptr, eta, n, bytes, out); }
void emu_sample_ntt(size_t n, const uint8_t* seeds, uint16_t* out) { sample_ntt_launch(nullptr, n, seeds, out); }
int emu_prf(int eta, size_t n, const uint8_t* in33, uint8_t* out) { return prf_launch(nullptr, eta, n, in33, out); }
int emu_hash(int kind, size_t n, const uint8_t* msg, unsigned len, size_t stride, uint8_t* out) {
    return hash_launch(nullptr, kind, n, msg, len, stride, out);
}
}
