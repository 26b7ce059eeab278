// Timing-only ablation switches of the MFMA kernels (tools/ablate_conv.py): every fork in the kernels is an
// `if constexpr (abl::<flag>)`, so BOTH sides are compiled and type-checked in every build, and this header is the only place
// that looks at the -DSHM_ABL_* macros.  In the product build every flag is false and the forks fold away; an ablated build
// computes WRONG results by design (it exists to time a kernel with one ingredient removed) and exports the same symbols
// (tools/ablate_conv.py asserts that).
#pragma once

namespace abl {
#define SHM_ABL_FLAG(name, macro) constexpr bool name = macro
#ifdef SHM_ABL_FIXADDR
SHM_ABL_FLAG(fixaddr, true);      // DMA tap GEMM: constant operand addresses, no per-step address work
#else
SHM_ABL_FLAG(fixaddr, false);
#endif
#ifdef SHM_ABL_SAMELINE
SHM_ABL_FLAG(sameline, true);     // every operand load hits the same cache lines (no HBM / L2 traffic)
#else
SHM_ABL_FLAG(sameline, false);
#endif
#ifdef SHM_ABL_NODMA
SHM_ABL_FLAG(nodma, true);        // no in-loop LDS-DMA (the prologue's stages are reused)
#else
SHM_ABL_FLAG(nodma, false);
#endif
#ifdef SHM_ABL_NOLDS
SHM_ABL_FLAG(nolds, true);        // no LDS fragment reads (operands from registers)
#else
SHM_ABL_FLAG(nolds, false);
#endif
#ifdef SHM_ABL_NOMFMA
SHM_ABL_FLAG(nomfma, true);       // no MFMAs
#else
SHM_ABL_FLAG(nomfma, false);
#endif
#ifdef SHM_ABL_NOSTORE
SHM_ABL_FLAG(nostore, true);      // no output stores
#else
SHM_ABL_FLAG(nostore, false);
#endif
#ifdef SHM_ABL_NOEPI
SHM_ABL_FLAG(noepi, true);        // weights-in-registers kernel: no epilogue at all
#else
SHM_ABL_FLAG(noepi, false);
#endif
#ifdef SHM_ABL_NOBAR
SHM_ABL_FLAG(nobar, true);        // weight gradient: no barriers
#else
SHM_ABL_FLAG(nobar, false);
#endif
#ifdef SHM_ABL_NOLOAD
SHM_ABL_FLAG(noload, true);       // weight gradient: no global loads
#else
SHM_ABL_FLAG(noload, false);
#endif
#ifdef SHM_WREG_PRIO
SHM_ABL_FLAG(wreg_prio, true);    // weights-in-registers kernel: s_setprio 1 around the MFMA loop (measured: no effect)
#else
SHM_ABL_FLAG(wreg_prio, false);
#endif
#ifdef SHM_ABL_STAMP
SHM_ABL_FLAG(stamp, true);        // tapgemm_wreg16_bf16_kernel: s_memtime stamps at the phase boundaries of a patch; one block dumps its per-wave totals over the bias vector
#else
SHM_ABL_FLAG(stamp, false);
#endif
#undef SHM_ABL_FLAG
}  // namespace abl
