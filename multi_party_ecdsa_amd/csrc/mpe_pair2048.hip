// Translation unit of the N-adic pair engine for 2048-bit moduli (arithmetic modulo their squares); see mpe_pairexp.h.
// Kept apart from mpe_lib.hip only so that the three units compile in parallel.
#include "mpe_pairexp.h"

namespace mpe {

int pairset_create_2048(int count, const uint32_t* d_moduli, mpe_pairset** out, hipStream_t st) {
  return pairset_create_impl<Cfg2048>(count, d_moduli, out, st);
}
int pair_modexp_2048(mpe_ctx* ctx, const mpe_pairset* ps, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                     Rows base2, Rows exps2, int exp2_words, int half, uint32_t* out, hipStream_t st, int public_exp) {
  // A launch lasts as long as ONE exponentiation however few there are.  When the batch fills less than half of
  // the resident groups, spread every integer over twice the lanes (9 limbs per lane): the same limbs, the same
  // per-modulus constants, about half the latency.
  using Wide = Cfg<2048, MPE_W, MPE_L / 2, 8>;
  static_assert(Wide::K == Cfg2048::K, "the two layouts share the limb arrays");
  const long resident = (long)ctx->cus * ctx->modexp_waves_per_cu * Cfg2048::GROUPS / ctx->device_share;   // this context's share of the chip
  // ... and a really small batch (a sixteenth of the resident groups) over four times the lanes (5 limbs per lane: the same
  // constants, zero-padded).  Measured (profiles/r03/xwide_sweep.json): -6 % per batch at 256 sessions, nothing at 1 024 and
  // +13 % when the threshold lets mid-size launches take it: with 10 MACs per step the quotient-digit dependency chain
  // (mad -> mul_lo -> three DPP moves -> mad) is no longer hidden, so the layout only pays while the chip is nearly empty.
  using XWide = Cfg<2048, MPE_W, 5, 16>;
  const int xdiv = ctx->xwide_div;                            // MPE_XWIDE_DIV, read when the context was created; 0 switches the layout off
  if (ctx->adaptive_lanes && xdiv > 0 && (long)xdiv * batch <= resident)
    return pair_modexp_impl<XWide>(ctx, ps, batch, mod_sel, base, exps, exp_words, base2, exps2, exp2_words, half, out, st, public_exp);
  if (ctx->adaptive_lanes && (long)ctx->wide_div * batch <= resident)
    return pair_modexp_impl<Wide>(ctx, ps, batch, mod_sel, base, exps, exp_words, base2, exps2, exp2_words, half, out, st, public_exp);
  return pair_modexp_impl<Cfg2048>(ctx, ps, batch, mod_sel, base, exps, exp_words, base2, exps2, exp2_words, half, out, st, public_exp);
}

}  // namespace mpe

#ifdef MPE_WAVE_TRACE
extern "C" int mpe_wave_trace_arm_2048(void* d_buf, unsigned cap) { return mpe::wave_trace_arm_impl(d_buf, cap); }
extern "C" int mpe_wave_trace_count_2048(unsigned* n) { return mpe::wave_trace_count_impl(n); }
#endif
