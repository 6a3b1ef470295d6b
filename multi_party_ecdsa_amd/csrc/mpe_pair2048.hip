// Translation unit of the N-adic pair engine for 2048-bit moduli (arithmetic modulo their squares); see mpe_pairexp.h.
// Kept apart from mpe_lib.hip only so that the three units compile in parallel.
#include "mpe_pairexp.h"

namespace mpe {

int pairset_create_2048(int count, const uint32_t* d_moduli, mpe_pairset** out, hipStream_t st) {
  return pairset_create_impl<Cfg2048>(count, d_moduli, out, st);
}
int pair_modexp_2048(mpe_ctx* ctx, const mpe_pairset* ps, int batch, Rows mod_sel, Rows base, Rows exps, int exp_words,
                     Rows base2, Rows exps2, int exp2_words, int half, uint32_t* out, hipStream_t st) {
  return pair_modexp_impl<Cfg2048>(ctx, ps, batch, mod_sel, base, exps, exp_words, base2, exps2, exp2_words, half, out, st);
}

}  // namespace mpe
