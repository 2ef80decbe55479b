/*
 * mlkem_compat.h — ABI declarations of the ml_kem.h-compatible drop-in shim (libml_kem.so).
 *
 * A program written against the reference's ml_kem.h (rsjahnige/CRYSTALS-Kyber) keeps including the
 * reference's own header and simply links libml_kem.so instead of ml_kem.o + sha3.o; this header only
 * restates the binary interface the shim implements so that the shim and its tests compile without the
 * reference tree:
 *
 *   symbol        reference declaration      replaced definition
 *   ml_errno      ml_kem.h:26                ml_kem.c:16
 *   init          ml_kem.h:94                ml_kem.c:1363-1395
 *   KEM_KeyGen    ml_kem.h:68                ml_kem.c:1233-1252
 *   KEM_Encaps    ml_kem.h:76                ml_kem.c:1257-1305
 *   KEM_Decaps    ml_kem.h:83                ml_kem.c:1310-1359
 *
 * Layout facts that matter: the "byte" cell is a union of unsigned-int bit-fields and is
 * therefore 4 bytes wide, value in bits 0-7, upper 24 bits undefined on input (the shim never reads them)
 * and zero on output.  sizeof(struct PARAMS) = 20, sizeof(struct PKE) = 24, sizeof(struct KEM) = 144.
 * Every returned array is malloc()ed by the callee and free()d by the caller, as in the reference.
 */
#ifndef MLKEM_COMPAT_H
#define MLKEM_COMPAT_H

#ifdef __cplusplus
extern "C" {
#endif

extern int ml_errno;

union byte {
    unsigned int s : 7;
    unsigned int e : 8;
};

struct PARAMS {
    union byte k, n1, n2, du, dv;
};

struct PKE {
    union byte* ek;
    union byte* dk;
    unsigned int ek_len, dk_len;
};

struct KEM {
    union byte K[32];
    union byte* c;
    unsigned int c_len;
};

enum ML_KEM { ML_KEM_512 = 512, ML_KEM_768 = 768, ML_KEM_1024 = 1024 };

const struct PARAMS init(enum ML_KEM param_set);

/* ---- sha3.h surface re-exported through ml_kem.h:10 --------------------------------------------
 *   symbol   reference declaration   replaced definition
 *   h2b      sha3.h:28               sha3.c:329-359
 *   b2h      sha3.h:35               sha3.c:367-396
 *   sha3_b   sha3.h:44               sha3.c:408-436 (+ Sponge :257, pad :226, Keccak_f :207)
 *   sha3_h   sha3.h:54               sha3.c:443-457
 *   sha3_s   sha3.h:64               sha3.c:465-494
 * `union bit` / `union hex` are 4-byte cells like `union byte`.  Every result is malloc()ed for the caller.
 * Capacities c with rate (1600 - c)/8 in {72, 104, 136, 144, 168} bytes are supported (all SHA-3 / SHAKE
 * instances); the reference's padding bug for bit lengths = r-2 mod r is NOT reproduced. */
union bit {
    unsigned int b : 1;
};
union hex {
    unsigned int d : 4;
};
union bit* h2b(const union hex* H, unsigned int m, unsigned int n);
union hex* b2h(const union bit* S, unsigned int n);
union bit* sha3_b(const union bit* bstr, unsigned int n, unsigned int d, unsigned int c, union bit sfx[4]);
union hex* sha3_h(const union hex* hstr, unsigned int m, unsigned int d, unsigned int c, union bit sfx[4]);
unsigned char* sha3_s(const char* cstr, unsigned int m, unsigned int d, unsigned int c, union bit sfx[4]);
/* ---- primitives the reference's objects export without declaring them (`nm ml_kem.o` shows T SampleNTT,
 * SamplePolyCBD, NTT, InverseNTT; its own test programs Test_Archive/SampleNTT_test06.c, SampleCBD_test07.c and
 * NTT_test08.c call them).  `union integer` (ml_kem.c:20-23) is a 4-byte cell like `union byte`: the coefficient is
 * the 12-bit field `t` in bits 0-11; inputs are read through `t` (mod 2^12), outputs are canonical in [0, q).
 *   symbol          replaced definition     in -> out (malloc()ed for the caller)
 *   SampleNTT       ml_kem.c:189-245        B[34] cells -> 256 cells     (like the reference, B[32].e and B[33].e are
 *                                           incremented once per 279-triple retry, probability < 2^-300: mlkem_sample_ntt_retries)
 *   SamplePolyCBD   ml_kem.c:253-275        B[64 n] cells, n = eta in {2, 3} -> 256 cells
 *   NTT             ml_kem.c:287-329        256 cells -> 256 cells
 *   InverseNTT      ml_kem.c:336-384        256 cells -> 256 cells */
union integer {
    unsigned int t : 12;
    unsigned int l : 24;
};
union integer* SampleNTT(union byte* B);
union integer* SamplePolyCBD(const union byte* B, unsigned int n);
union integer* NTT(const union integer* f);
union integer* InverseNTT(const union integer* fh);
struct PKE KEM_KeyGen(const struct PARAMS* params);
struct KEM KEM_Encaps(const struct PARAMS* params, const union byte* ek, unsigned int ek_len);
union byte* KEM_Decaps(const struct PARAMS* params, const union byte* dk, unsigned int dk_len, const union byte* c,
                       unsigned int c_len);

#ifdef __cplusplus
}
#endif
#endif
