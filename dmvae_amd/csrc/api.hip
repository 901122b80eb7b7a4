// C-ABI plumbing: thread-local error string + version.
#include "common.h"
#include "dmvae_hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void dmvae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dmvae_last_error(void) { return g_err; }
extern "C" int dmvae_abi_version(void) { return 8; }   // 8: dmvae_reparam_kl_* (the reparameterise hook + posterior-form KL); 7: the XCD-placed grouped weight-gradient launch (dmvae_linear_wgrad_grouped_plan / _xcd); 6: the whole-stack DiT backward + batched per-sample Linears + batched weight transposes; 5: the shortcut-in-GroupNorm entry points (dmvae_groupnorm_*_short); 2: dmvae_conv_desc gained w_layout; 3: dmvae_pack_entry + the batched pack / Linear GEMM entry points; 4: the decoder-tail entry points (dmvae_norm_conv_out_*)
