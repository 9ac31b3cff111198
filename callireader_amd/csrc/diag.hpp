// Diagnostic builds, quarantined (round-4 verdict, item 8).
//
// The knock-out / knock-in / poison / break-wait / stamp blocks inside the product kernels (gemm256.hip, gemm_skinny.hip, attention_vit.hip) are cost-structure
// and hazard-screen tools: several of them give WRONG RESULTS by design.  Every translation unit that has such blocks includes this header, and this header is
// the only door to them:
//   * a diagnostic macro without -DCR_DIAG_BUILD is a compile error -- callireader_amd/build.py (the product build, __graft_entry__.build()) never defines
//     it and refuses -D flags of its own; only scripts/build_variant.py does, and it writes its libraries under ab/, never over the product's;
//   * a translation unit built with any of them registers the macro names at load time: cr_build_flags() returns them ("" for the product build), and
//     callireader_amd/_binding.py refuses to load a library that reports any unless CR_HIP_LIB names that library explicitly;
//   * scripts/build_variant.py also hashes the flags into cr_build_id(), so a variant never carries the product's id.
#pragma once

// (wrong results by design: CR_BREAK_WAIT, CR_KO_STORE, CR_KO_WCONTIG, CR_KO_AROWS, CR_KO_EPI, CR_KO_XFRAG, CR_KO_VIT_SOFTMAX, CR_KO_VIT_MFMA; right results, other cost or extra
//  output: CR_POISON, CR_KI_VALU, CR_DIAG_STAMPS, CR_TILE_GM -- the binding refuses all of them alike)

#ifdef CR_BREAK_WAIT
#define CR_DIAG_S1 "CR_BREAK_WAIT "
#else
#define CR_DIAG_S1 ""
#endif
#ifdef CR_KO_STORE
#define CR_DIAG_S2 "CR_KO_STORE "
#else
#define CR_DIAG_S2 ""
#endif
#ifdef CR_KO_WCONTIG
#define CR_DIAG_S3 "CR_KO_WCONTIG "
#else
#define CR_DIAG_S3 ""
#endif
#ifdef CR_KO_AROWS
#define CR_DIAG_S4 "CR_KO_AROWS "
#else
#define CR_DIAG_S4 ""
#endif
#ifdef CR_KO_EPI
#define CR_DIAG_S5 "CR_KO_EPI "
#else
#define CR_DIAG_S5 ""
#endif
#ifdef CR_KO_XFRAG
#define CR_DIAG_S6 "CR_KO_XFRAG "
#else
#define CR_DIAG_S6 ""
#endif
#ifdef CR_TILE_GM
#define CR_DIAG_S7 "CR_TILE_GM "
#else
#define CR_DIAG_S7 ""      /* (round 4's CR_KO_W8CONTIG became the e4m3 copies' decode layout) */
#endif
#ifdef CR_POISON
#define CR_DIAG_S8 "CR_POISON "
#else
#define CR_DIAG_S8 ""
#endif
#ifdef CR_KI_VALU
#define CR_DIAG_S9 "CR_KI_VALU "
#else
#define CR_DIAG_S9 ""
#endif
#ifdef CR_DIAG_STAMPS
#define CR_DIAG_S10 "CR_DIAG_STAMPS "
#else
#define CR_DIAG_S10 ""
#endif
#ifdef CR_KO_VIT_SOFTMAX
#define CR_DIAG_S11 "CR_KO_VIT_SOFTMAX "
#else
#define CR_DIAG_S11 ""
#endif
#ifdef CR_KO_VIT_MFMA
#define CR_DIAG_S12 "CR_KO_VIT_MFMA "
#else
#define CR_DIAG_S12 ""
#endif

#if defined(CR_BREAK_WAIT) || defined(CR_KO_STORE) || defined(CR_KO_WCONTIG) || defined(CR_KO_AROWS) || defined(CR_KO_EPI) || defined(CR_KO_XFRAG) || \
    defined(CR_POISON) || defined(CR_KI_VALU) || defined(CR_DIAG_STAMPS) || defined(CR_KO_VIT_SOFTMAX) || defined(CR_KO_VIT_MFMA) || defined(CR_TILE_GM)
#ifndef CR_DIAG_BUILD
#error "diagnostic macros (CR_KO_*, CR_KI_VALU, CR_POISON, CR_BREAK_WAIT, CR_DIAG_STAMPS) are only available through scripts/build_variant.py (-DCR_DIAG_BUILD): csrc/diag.hpp"
#endif
extern "C" int cr_diag_register(const char* flags);
namespace {
struct CrDiagRegistration {
    CrDiagRegistration() { cr_diag_register(CR_DIAG_S1 CR_DIAG_S2 CR_DIAG_S3 CR_DIAG_S4 CR_DIAG_S5 CR_DIAG_S6 CR_DIAG_S7 CR_DIAG_S8 CR_DIAG_S9 CR_DIAG_S10 CR_DIAG_S11 CR_DIAG_S12); }
};
static CrDiagRegistration cr_diag_registration_;
}  // namespace
#endif
