// Launch wrappers and device-visible descriptors of the streaming path (k_stream.hip).
// Reference: core/moonshine-streaming-model.cpp (driver) over the five graphs of
// language-bindings/python/src/moonshine_voice/lora/export.py; see stream_engine.h for the data layout.
#pragma once

#include "kernels.h"

namespace msh {

// contiguous device-to-device copy, 16-byte granularity
struct StreamSeg {
  const void* src;
  void* dst;
  long bytes;
};

// new audio of one stream in one process_audio call
struct FrameJob {
  long audio_off;  // first sample in the staged audio buffer
  int n_frames;    // new 80-sample frames (multiple of 4)
  int row0;        // packed row of the first new frame
};

// per-stream decoder bookkeeping that lives on the device
struct SlotDev {
  int mem_len;     // memory frames = cross-attention keys
  int cache_len;   // self-attention cache length
  int count;       // tokens written by decode_full
  int finished;
  int max_tokens;
  int current;     // token the next step feeds
  int accepted;    // draft tokens accepted by the last verify
  int pad_;
};

// one stream's rows in a wide decoder pass: [BOS, draft...]
struct DecJob {
  int slot;
  int row0;
  int n_rows;
  int draft_off;  // into the flat draft array
  int draft_len;
  int max_tokens;
  int base;  // self-cache length before the pass
  int pad_;
};

// frames[row0 + f][0..95] = bf16(asinh(k * cmvn(audio frame f))) (cols 80..95 zero)
// (modeling_moonshine_streaming.py:70-88; K padded 80 -> 96 for the MFMA k-slices)
void stream_frames(const float* audio, const FrameJob* jobs, int n_jobs, int max_frames, float k, bf16_t* frames,
                   hipStream_t s);
void copy_segments(const StreamSeg* segs, int n, hipStream_t s);
// sliding-window encoder self-attention (no RoPE): row i attends rows [max(i-past, lo_i), min(i+future, hi_i-1)]
// of the packed stream (lora/export.py:113-127: both bounds inclusive); qkv [R,3D] bf16 -> out [R,D] bf16
// tile_row0 / n_tiles (optional): the first rows of the call's tiles of <= 16 consecutive rows of one stream, counted from
// each stream's first row -- with them (and head_dim % 8 == 0) the MFMA kernel runs, one wave per (tile, head)
void stream_enc_attention(const bf16_t* qkv, const int* row_lo, const int* row_hi, int R, int D, int heads, int past,
                          int future, bf16_t* out, hipStream_t s, const int* tile_row0 = nullptr, int n_tiles = 0);
// adapter input: out[i] = y32[rows[i]] + pos_emb[pos[i]]  (lora/export.py:141-144), as bf16 and fp32
void stream_adapter_in(const float* y32, const int* rows, const int* pos, int n, int D, const float* pos_emb,
                       bf16_t* out16, float* out32, hipStream_t s);
// tmp [n][L*2*D] (per layer: k row, v row) -> crossK / crossV [slot][L][D][Mcap] (transposed, keys contiguous)
// at memory index idx[i]
void stream_scatter_cross(const bf16_t* tmp, const int* slot, const int* idx, int n, int L, int D, int Mcap,
                          bf16_t* crossK, bf16_t* crossV, hipStream_t s);
void stream_embed(const int* tokens, int M, const float* embed, int D, float* H, hipStream_t s);
// append k / v of every row to the self cache [slot][L][Scap][D] at row_pos, then causal attention over [0, row_pos]
void stream_self_attention(const bf16_t* qkv, const int* row_slot, const int* row_pos, int M, int D, int heads, int layer,
                           int L, int Scap, bf16_t* cacheK, bf16_t* cacheV, bf16_t* out, hipStream_t s);
// same attention when k / v of the rows are already in the cache (written by the QKV GEMM epilogue): q [M,D] bf16.
// fm (here and below): `out` / `H` of the AR steps in the fragment-major layouts of kernels.h (fm16 / fm32, rows padded
// to 16) -- the operand layout of the FM decode GEMMs that consume them.  max_keys: an upper bound of row_pos + 1 over the
// rows when the caller has one (0 = none): up to 128 keys the one-round-trip AR kernel runs (same arithmetic and order)
void stream_self_attention_cached(const bf16_t* q, const int* row_slot, const int* row_pos, int M, int D, int heads,
                                  int layer, int L, int Scap, const bf16_t* cacheK, const bf16_t* cacheV, bf16_t* out,
                                  hipStream_t s, bool fm = false, int max_keys = 0);
// cross-attention of every row over its stream's memory (keys [0, slots[slot].mem_len))
// word timestamps: softmax probabilities of every (row, head) over the stream's memory frames -> out[row][layer][head][Ecap]
void stream_cross_probs(const bf16_t* q, const int* row_slot, const SlotDev* slots, int M, int D, int heads, int layer,
                        int L, int Mcap, const bf16_t* crossK, int Ecap, float* out, hipStream_t s);
void stream_cross_attention(const bf16_t* q, const int* row_slot, const SlotDev* slots, int M, int D, int heads,
                            int layer, int L, int Mcap, const bf16_t* crossK, const bf16_t* crossV, bf16_t* out,
                            hipStream_t s, bool fm = false, const int* row_mem = nullptr);
// The same for runs of consecutive rows of one stream (runs[i] = {first row, rows <= kCrossRunRows}): a run's rows share one
// pass over the stream's K / V.  Results equal stream_cross_attention's bit for bit.
constexpr int kCrossRunRows = 4;
void stream_cross_attention_runs(const bf16_t* q, const int* row_slot, const int2* runs, int n_runs, const SlotDev* slots, int D,
                                 int heads, int layer, int L, int Mcap, const bf16_t* crossK, const bf16_t* crossV, bf16_t* out,
                                 hipStream_t s);
bool stream_cross_attention_runs_supported(int D, int heads, int Mcap);
// Long runs (the verify pass: up to kCrossWideRows consecutive rows of one stream per run) on the matrix pipe: one workgroup per
// (run, head) streams the stream's K / V once.  P is rounded to bf16: results equal the kernels above to rounding, not bit for bit.
constexpr int kCrossWideRows = 128;
void stream_cross_attention_wide(const bf16_t* q, const int* row_slot, const int2* runs, int n_runs, const SlotDev* slots, int D,
                                 int heads, int layer, int L, int Mcap, const bf16_t* crossK, const bf16_t* crossV, bf16_t* out,
                                 hipStream_t s);
bool stream_cross_attention_wide_supported(int D, int heads, int Mcap);
// first-max argmax of every logits row (moonshine-streaming-model.cpp:1222-1232)
void stream_argmax(const float* logits, int M, int V, int* pred, hipStream_t s);
// speculative verify (moonshine-streaming-model.cpp:1304-1366): longest agreeing draft prefix, rollback of the
// self cache length, first continuation token; prepares row j of the step buffers (H, step_pos) for job j.
void stream_verify(const DecJob* jobs, int n_jobs, const int* pred, const int* draft, SlotDev* slots, int* result,
                   int result_stride, int eos, const float* embed, int D, float* H, int* step_pos, int* n_active,
                   hipStream_t s, bool fm = false);
// one auto-regressive step of decode_full's loop (moonshine-streaming-model.cpp:1271-1288)
void stream_advance(const DecJob* jobs, int n_jobs, const int* pred, SlotDev* slots, int* result, int result_stride,
                    int eos, const float* embed, int D, float* H, int* step_pos, int* n_active, hipStream_t s,
                    bool fm = false);
// the same step from the per-tile (max, lowest index) pairs of gemm_argmax_partials (kernels.h): pval / pidx [n_jobs][ntn]
void stream_advance_partials(const DecJob* jobs, int n_jobs, const float* pval, const int* pidx, int ntn, SlotDev* slots,
                             int* result, int result_stride, int eos, const float* embed, int D, float* H, int* step_pos,
                             int* n_active, hipStream_t s, bool fm = false);
// Contextual biasing (reference core/context-biaser.cpp:88-149): flat trie over token ids, children sorted by token.
struct BiasTrie {
  const int* child_off;    // [n_nodes + 1]
  const int* child_tok;
  const int* child_node;
  const int* depth;        // [n_nodes]
  const float* depth_bonus;  // [max_depth + 2]
  int n_nodes;
};
// Adds the bonuses to logits rows before the argmax.  The active trie nodes of a row are recomputed from the tokens
// that precede it (the walk of ContextBiaser::advance from the root over that prefix):
//   wide pass (jobs == nullptr): row r has prefix tokens[prefix[r].x .. prefix[r].x + prefix[r].y)
//   AR step   (jobs != nullptr): row j belongs to jobs[j].slot, prefix = result[slot][0 .. slots[slot].count);
//                                 finished streams are skipped
void stream_bias_rows(BiasTrie trie, const int2* prefix, const int* tokens, const DecJob* jobs, const SlotDev* slots,
                      const int* result, int result_stride, int rows, float* logits, int V, hipStream_t s);
// per entry (slot, mem_len or -1 = keep, cache_len or -1 = keep, nonzero = clear the decode_full fields)
void stream_slot_update(const int4* upd, int n, SlotDev* slots, hipStream_t s);
// slots[slot].cache_len += n for the rows of a plain wide pass (decode_tokens)
void stream_bump_cache(const DecJob* jobs, int n_jobs, SlotDev* slots, hipStream_t s);

}  // namespace msh
