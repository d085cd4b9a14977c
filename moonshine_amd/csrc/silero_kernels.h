// Launch wrappers of k_silero.hip: the Silero VAD network on the device, batched over hops (fp32 throughout).
#pragma once

#include "msh_common.h"

namespace msh {

// 16-bit PCM -> fp32 (x / 32768), n samples; src 8-byte, dst 16-byte aligned
void silero_pcm16_to_f32(const int16_t* src, float* dst, long n, hipStream_t s);
// frames [(hop * 4 + t)][256] from the flat audio buffer; hop_base[h] = index of the 64 context samples in front of hop h
void silero_frames(const float* audio, const long* hop_base, long n_hops, float* frames, hipStream_t s);
// |STFT|: frames x basis[258][256]^T -> stft_tmp [hops * 4][258] -> mag [hop][129][4]
void silero_stft_mag(const float* frames, const float* basis, long n_hops, float* stft_tmp, float* mag, hipStream_t s);
// Conv1d(k = 3, padding = 1, stride) + ReLU: in [hop][cin][tin] -> out [hop][cout][tout]; w_padded [cout][kpad] is the
// [cout][cin * 3] weight with zero columns up to kpad (a multiple of 16); cols = scratch [hops * tout][kpad]
void silero_conv_relu(const float* in, int cin, int tin, int stride, const float* w_padded, int kpad, const float* bias, int cout,
                      long n_hops, float* cols, float* out, hipStream_t s);
// gin [hop][512] = feat [hop][128] x w_ih[512][128]^T + bias_sum (b_ih + b_hh)
void silero_gate_inputs(const float* feat, const float* w_ih, const float* bias_sum, long n_hops, float* gin, hipStream_t s);
// the recurrence over each clip's hops (clip_hop0[c] .. clip_hop0[c + 1]), fresh state per clip; probs [hop]
void silero_lstm(const float* gin, const float* w_hh, const float* out_w, float out_b, const long* clip_hop0, int n_clips,
                 float* probs, hipStream_t s);

}  // namespace msh
