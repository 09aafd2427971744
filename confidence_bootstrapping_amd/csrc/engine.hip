// Host side of the MI355X docking engine: C ABI of include/cbdock.h.
// Owns weights (re-packed for the MFMA streams), the per-complex static data and the per-batch workspace, and
// sequences the kernels of one score-model forward pass / one reverse-diffusion step.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "host_util.h"
#include "kernels.h"

using namespace cbd;

static thread_local std::string g_err;
int cbd_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

namespace {

struct ConvLayerDev {
  int in_level = 0, out_level = 0, n_groups = 0;
  float* wstream[4] = {nullptr, nullptr, nullptr, nullptr};
  float* wstream_bf16[4] = {nullptr, nullptr, nullptr, nullptr};        // bf16-operand policy (tp_conv.hip::OpsBf16)
  float* wstream_x3[4] = {nullptr, nullptr, nullptr, nullptr};          // bf16x3 fp32-emulation policy (OpsBf16x3)
  float* w1sd[4] = {nullptr, nullptr, nullptr, nullptr};                // [2][32][96]: W1[:, 32:64]^T and W1[:, 64:96]^T (node parts of the first Linear)
  float *bn_scale = nullptr, *bn_mean = nullptr, *bn_bias = nullptr;   // [NODE_STRIDE]
};

struct MlpDev {   // 2-layer edge MLP pieces
  float *WgT = nullptr, *WbT = nullptr, *W1T = nullptr, *b0 = nullptr, *b1 = nullptr, *offset = nullptr;
  float coeff = 0.f;
};

}  // namespace

struct cbd_engine {
  cbd_config cfg{};
  std::map<std::string, HostTensor> host_w;
  bool weights_ready = false, complex_ready = false;
  DevPool wpool, cpool, bpool;   // weights / complex / batch workspace
  // "async_setup" = 1: cbd_set_complex works on `setup` (a stream of its own) and waits only for THIS engine's last launches (`ev_last`,
  // recorded by the sampling entry points) instead of synchronising the device and using the default stream -- the set-up of the next
  // group of complexes then overlaps the step loop of the current one (sampling.py).  `sync_all`: a launch through another entry point
  // (cbd_score, cbd_modify_conformer, ...) since the last set-up: that set-up synchronises the device as before.
  hipStream_t setup = nullptr;
  hipEvent_t ev_last = nullptr;
  bool last_used = false, sync_all = false;     // a fresh engine has launched nothing
  int async_setup = 0;

  // ---- weights on device
  ConvLayerDev rec_emb[3], lig_emb[3], conv[5];
  MlpDev m_lig_edge, m_cross, m_rec_edge, m_final_edge, m_center;
  StepWeights sw{};
  CenterHead ch{};
  BondHead bh{};
  float* bond_stream = nullptr;     // tor_bond_conv FCBlock as an MFMA tile stream (bond_conv_kernel)
  float *rec_emb_table = nullptr, *rec_node_w = nullptr, *rec_node_b = nullptr;
  std::vector<float> lig_node_w_host;   // additional_features_embedder [32][64]
  std::vector<std::vector<float>> lig_emb_tables;

  // ---- complex
  GraphStatic gs{};
  int cap_ll_per_sample = 0;
  float *lig_static32 = nullptr, *rec_static = nullptr, *rr_attr0 = nullptr, *rr_attr_t = nullptr;
  int *rr_src = nullptr, *rr_dst = nullptr, *rr_aidx = nullptr;
  float* rr_vec = nullptr;
  int* rr_count_dev = nullptr;
  float *d_rec_x = nullptr, *d_vec0 = nullptr, *d_dist0 = nullptr;
  int *d_src0 = nullptr, *d_dst0 = nullptr, *d_ident = nullptr, *d_deg0 = nullptr;
  int use_bf16 = 0;                 // operand policy of the tensor-product kernel: 0 exact fp32 MFMA, 1 bf16 ("bf16" option),
                                    // 2 fp32 emulated by three bf16 planes ("f32_split" option)
  // hipGraph of the step loop (optional).  The first engine of a co-scheduled group owns the instantiated graphs of the group
  // (key: schedule, batch sizes, engines and their complex generations); every engine owns its staging buffers.
  bool use_graph = false;
  struct GraphEntry { std::string key; hipGraphExec_t exec; size_t ev_lo, ev_hi; bool replayed; };   // [ev_lo, ev_hi): timing events recorded by its nodes
  std::vector<GraphEntry> graphs;
  float *g_pos = nullptr, *g_ztr = nullptr, *g_zrot = nullptr, *g_ztor = nullptr;
  int g_S_cap = 0;
  // Generation stamp of everything a captured launch of this engine depends on (complex, staging buffers, weights).  Drawn from ONE
  // process-wide monotonic counter (next_gen), never from a per-engine 0: the graphs of a co-scheduled group live on its first engine
  // and are keyed by (engine address, generation) of every member, so a partner that is destroyed and re-created at the same heap
  // address must not reach a generation an old key already holds (its kernel arguments point at freed device buffers).
  uint64_t complex_gen = next_gen();
  static uint64_t next_gen() { static std::atomic<uint64_t> g{1}; return g.fetch_add(1, std::memory_order_relaxed); }
  // device-resident description of the pose batch of the current call (kernels.h::PoseBatch) and its host image
  PoseBatch desc_h{};
  PoseBatch* desc_dev = nullptr;
  // per edge group (ll, lr, rr, rl, rr0 = single-copy receptor graph): pieces of the deterministic segmented reduction
  float *fsum[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, *lsum[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* racc[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float *fsum_x[2] = {nullptr, nullptr}, *lsum_x[2] = {nullptr, nullptr}, *racc_x[2] = {nullptr, nullptr};   // extra slices of ll (embedding layers)
  // bf16 role split ("bf16_roles" option): the cross / receptor groups (1 lr, 2 rr, 3 rl) run as three virtual slices per layer --
  // 0e tiles [0, 19), 0e tiles [19, 38), vector blocks; the second 0e slice writes piece buffers of its own, laid out exactly like
  // the group's and `piece_b_off[g]` floats behind them (first_sum, last_sum and run_acc alike)
  // EXPERIMENT (lost: 217 -> 163 poses/s): compiled only into the diagnostic library (-DCBD_EXPERIMENTS,
  // tools/diag_lib.py, experiments/csrc/tp_conv_bf16p.hip); in the product library `bf16_roles` is the constant 0 and the option is refused
#ifdef CBD_EXPERIMENTS
  long long piece_b_off[4] = {0, 0, 0, 0};
  int bf16_roles = 0;      // 1: the slices run through the streaming kernel, 2: the 0e slices through the LDS-resident kernel (tp_conv_bf16p.hip)
#else
  static constexpr int bf16_roles = 0;
#endif
  // 1: the 74 -> 74 layers of the bf16 policy run through the register-stationary kernel (tp_conv_bf16s.hip; "bf16_stationary" option / CBD_BF16_STATIONARY;
  // the default since the end of round 5: 2 .. 4 % faster than the streaming kernel on C2 and C4, profiles/r05_c_*); 0: through the streaming
  // kernel.  Ignored under the role split.
  int bf16_stat = 1;
  int n_cus = 256;
  int *rr_start = nullptr, *rr_cnt = nullptr;   // [max_batch*Nr] CSR ranges of the batched receptor edges
  int *rr0_start = nullptr;                     // [Nr] CSR starts of the single-copy receptor edges
  hipStream_t own = nullptr;        // used instead of the legacy default stream for graph capture (which cannot be captured)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  hipStream_t side = nullptr;       // forked stream for work that only depends on the diffusion time
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  unsigned long long* stamps_dev = nullptr;   // diagnostic (CBD_CONV_VARIANT=8)
  unsigned long long* stats_dev = nullptr;   // [4] edge-layer visits: ll (embedding layers), joint conv layers, forwards

  // ---- batch workspace
  GraphDyn gd{};
  float *X0 = nullptr, *X1 = nullptr;
  float* proj[2 * 4 + 2] = {};   // node projections of the first Linear: (src, dst) x up to 4 FCBlocks of a layer, + one pair for the side stream
  float *ll_attr = nullptr, *lr_attr = nullptr;
  StepVectors sv{};
  float* sigma_emb_dev = nullptr;   // [S_max][64]: sigma_emb | sigma_emb_t of every step
  int sigma_cap = 0;
  float *tr_out = nullptr, *rot_out = nullptr, *tor_out = nullptr, *dbg_global = nullptr, *dbg_torfeat = nullptr;
  float* center_msg = nullptr;
  int *tor_nb = nullptr, *tor_nb_cnt = nullptr;
  int n_nodes_cap = 0;
  int last_B = 0;
  std::map<std::string, std::pair<const float*, size_t>> dbg;   // name -> (device ptr, count), valid after cbd_score
  std::vector<std::pair<std::string, std::vector<float>>> dbg_snap;   // snapshots of ping-pong buffers
  bool keep_debug = false;

  // ---- timing of the dominant kernel
  bool timing = false;
  typedef std::vector<std::pair<hipEvent_t, hipEvent_t>> EvPool;
  EvPool ev_pool;       // pairs of the eager launches since the last collection
  size_t ev_used = 0;
  EvPool gev_pool;      // pairs recorded by the nodes of the cached graphs (GraphEntry::ev_lo/ev_hi index this pool)
  size_t gev_used = 0;
  double t_total_ms = 0;
  int64_t t_n = 0;
};

static void fill_static_desc(cbd_engine* e);
static void drop_graphs(cbd_engine* e) {
  for (auto& g : e->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  e->gev_used = 0;   // the event pairs their nodes recorded into are free again
  e->graphs.clear();
  e->complex_gen = cbd_engine::next_gen();
}

// ======================================================================================================== weights
static const HostTensor* find_w(cbd_engine* e, const std::string& k) {
  auto it = e->host_w.find(k);
  return it == e->host_w.end() ? nullptr : &it->second;
}

static int need(cbd_engine* e, const std::string& k, std::initializer_list<int64_t> shape, const HostTensor** out) {
  const HostTensor* t = find_w(e, k);
  if (!t) return fail(CBD_ERR_WEIGHT, "missing tensor '%s' (load_state_dict strict=True)", k.c_str());
  if (t->shape != std::vector<int64_t>(shape)) return fail(CBD_ERR_WEIGHT, "tensor '%s' has an unexpected shape", k.c_str());
  *out = t;
  return 0;
}

// Re-pack one FCBlock (Linear 96->96, ReLU, Linear 96->W) into the tile stream consumed by tp_conv_kernel.
// See the layout notes at the top of tp_conv.hip.  W2's k order follows the C/D register layout of the first GEMM.
// Row description of every 32-row weight tile of a layer: which row of W1 / W2 (or none) each MFMA row carries and the
// folded coefficient (1/sqrt(fan_in), sqrt(3) of sh_1, 1/sqrt(2) of the cross paths).  Shared by the fp32 and bf16 packers.
struct TileRow { int row; float scale; };          // row < 0: zero row
struct TileRows { std::vector<TileRow> rows; int ntiles; };   // rows[T*32 + r]; tiles 0..2 index W1/b1, the rest W2/b2

static TileRows conv_tile_rows(int IN, int OUT, bool merged = false) {
  const ConvShape S = conv_shape(IN, OUT, merged);
  TileRows tr;
  tr.ntiles = S.ntiles;
  tr.rows.assign((size_t)S.ntiles * 32, TileRow{-1, 0.f});
  int T = 0;
  for (int m = 0; m < 3; ++m, ++T)
    for (int r = 0; r < 32; ++r) tr.rows[(size_t)T * 32 + r] = TileRow{32 * m + r, 1.0f};
  const float s3 = std::sqrt(3.0f), s15 = std::sqrt(1.5f);
  for (int i = 0; i < S.fan0e; ++i, ++T)
    for (int r = 0; r < 32; ++r) tr.rows[(size_t)T * 32 + r] = TileRow{i * NS + r, 1.0f / std::sqrt((float)S.fan0e)};
  // A block's tiles: its mids [0, n_mids) five per tile (weights at `off`, 1/sqrt(fan) folded), then -- in the free slots of its last
  // tile -- `guest_n` mids [guest_lo, ..) of ANOTHER block (weights at guest_off, that block's fan: ConvShape::vmerged).
  auto vec_tiles = [&](int off, int fan, int n_mids, auto mid_factor, int guest_off, int guest_fan, int guest_lo, int guest_n, auto guest_factor) {
    const int ntile = (n_mids + VEC_TILE_I - 1) / VEC_TILE_I;
    for (int t = 0; t < ntile; ++t, ++T)
      for (int r = 0; r < 32; ++r) {
        // row r = (reg&3) + 8*(reg>>2) + 4*hf  <->  reg = (r&3) + 4*(r>>3), hf = (r>>2)&1
        const int reg = (r & 3) + 4 * (r >> 3), hf = (r >> 2) & 1;
        const int i = VEC_TILE_I * t + reg / 3, o = 3 * hf + reg % 3;
        if (reg >= 15) continue;
        if (i < n_mids) tr.rows[(size_t)T * 32 + r] = TileRow{off + i * NV + o, mid_factor(i) / std::sqrt((float)fan)};
        else if (i - n_mids < guest_n) {
          const int gi = guest_lo + (i - n_mids);
          tr.rows[(size_t)T * 32 + r] = TileRow{guest_off + gi * NV + o, guest_factor(gi) / std::sqrt((float)guest_fan)};
        }
      }
  };
  auto none = [](int) { return 0.f; };
  auto f1o = [&](int i) { return i < NS ? s3 : (i < NS + S.n1o ? 1.0f : s15); };
  auto f1e = [&](int i) { return i < S.n1o ? s15 : (i < S.n1o + S.n1e ? 1.0f : s3); };
  const int off1o = S.fan0e * NS, off1e = off1o + S.fan1o * NV, off0o = off1e + S.fan1e * NV;
  const int own1e = S.vmerged ? VEC_TILE_I * (S.t1e - 1) : S.fan1e;       // merged: block 1e stops one tile early
  vec_tiles(off1o, S.fan1o, S.fan1o, f1o, 0, 1, 0, 0, none);
  if (OUT >= 2) vec_tiles(off1e, S.fan1e, own1e, f1e, 0, 1, 0, 0, none);
  if (OUT >= 3) vec_tiles(off0o, S.fan0o, S.fan0o, [&](int) { return 1.0f; }, off1e, S.fan1e, own1e, S.fan1e - own1e, f1e);
  return tr;
}

static std::vector<float> pack_rows_f32(const TileRows& tr, const float* W1, const float* b1, const float* W2, const float* b2);
static std::vector<float> pack_conv_stream(int IN, int OUT, const float* W1, const float* b1, const float* W2, const float* b2,
                                           bool merged = false) {
  return pack_rows_f32(conv_tile_rows(IN, OUT, merged), W1, b1, W2, b2);
}

// tor_bond_conv (e3nn FCTP with two live paths): 3 tiles of W1, then one tile per mid index u of path A (1o x T1 -> 32x0e,
// weight block [6][32] at offset 0) and of path B (1e x T1 -> 32x0o, offset 192); constants live in the kernel's mids.
static TileRows bond_tile_rows() {
  TileRows tr;
  tr.ntiles = BOND_CONV_TILES;
  tr.rows.assign((size_t)tr.ntiles * 32, TileRow{-1, 0.f});
  for (int m = 0; m < 3; ++m)
    for (int r = 0; r < 32; ++r) tr.rows[(size_t)m * 32 + r] = TileRow{32 * m + r, 1.0f};
  for (int q = 0; q < 12; ++q)
    for (int r = 0; r < 32; ++r) tr.rows[(size_t)(3 + q) * 32 + r] = TileRow{(q < 6 ? q * 32 : 192 + (q - 6) * 32) + r, 1.0f};
  return tr;
}

static std::vector<float> pack_rows_f32(const TileRows& tr, const float* W1, const float* b1, const float* W2, const float* b2) {
  std::vector<float> out(conv_stream_floats(tr.ntiles), 0.f);
  float* const bias_tab = out.data() + (size_t)(tr.ntiles + 1) * TILE_W_FLOATS;
  auto widx = [](int s, int lane) { return ((s >> 2) * 64 + lane) * 4 + (s & 3); };
  auto kperm = [](int s, int h) { return 32 * (s / 16) + ((s % 16) & 3) + 8 * ((s % 16) >> 2) + 4 * h; };
  for (int T = 0; T < tr.ntiles; ++T) {
    float* tile = out.data() + (size_t)T * TILE_W_FLOATS;
    const bool first = T < 3;
    for (int s = 0; s < KSTEPS; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        const TileRow& R = tr.rows[(size_t)T * 32 + r];
        if (R.row < 0) { tile[widx(s, lane)] = 0.f; continue; }
        // first Linear: lane half h holds input columns 16h..16h+15 of each 32-wide part; second Linear: C/D register order
        const int k = first ? 32 * (s / 16) + 16 * h + (s % 16) : kperm(s, h);
        tile[widx(s, lane)] = R.scale * (first ? W1 : W2)[(size_t)R.row * KDIM + k];
      }
    for (int r = 0; r < 32; ++r) {
      const TileRow& R = tr.rows[(size_t)T * 32 + r];
      bias_tab[(size_t)T * 32 + r] = R.row < 0 ? 0.f : R.scale * (first ? b1 : b2)[R.row];
    }
  }
  return out;
}

// bf16 stream of tp_conv.hip::OpsBf16: (ntiles + 1) tiles of [6 k-steps][64 lanes][8 bf16] (6 KB), then the fp32 bias table.
static uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static std::vector<float> pack_conv_stream_bf16(int IN, int OUT, const float* W1, const float* b1, const float* W2, const float* b2) {
  // stream of tp_conv_bf16.hip: (ntiles + 1) tiles of [6 k-steps][64 lanes][8 bf16] (6 KB) carrying the 96 input columns, then the fp32
  // bias rows [ntiles + 1][32] (the kernel feeds a tile's bias as the C operand of its first MFMA pair; row ntiles is the zero tile's)
  const TileRows tr = conv_tile_rows(IN, OUT, true);      // merged vector tails, as the fp32 inference stream
  constexpr int NQ = KDIM / 16;
  constexpr int TILE_BF16 = NQ * 64 * 8;   // 3072 bf16 = 6 KB
  const size_t tile_floats = (size_t)(tr.ntiles + 1) * TILE_BF16 / 2;
  // ... then the bias rows of the 0e tiles once more as a [32 outputs x 48 mids] bf16 MFMA tile (3 fragments of [64 lanes][8 bf16]):
  // the kernel adds sum_i b_i m_i of the 0e block with one small matrix product instead of feeding a bias to each of its tiles
  const size_t b0e_floats = 3 * 64 * 8 / 2;
  std::vector<float> out(tile_floats + (size_t)(tr.ntiles + 1) * 32 + b0e_floats, 0.f);
  uint16_t* const w = reinterpret_cast<uint16_t*>(out.data());
  float* const bias = out.data() + tile_floats;
  for (int T = 0; T < tr.ntiles; ++T) {
    uint16_t* tile = w + (size_t)T * TILE_BF16;
    const bool first = T < 3;
    for (int q = 0; q < NQ; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        const TileRow& R = tr.rows[(size_t)T * 32 + r];
        for (int j = 0; j < 8; ++j) {
          float v = 0.f;
          if (R.row >= 0) {
            // first Linear: k-step q = 2*part + sub covers input columns 16h + 8sub + j of the 32-wide part;
            // second Linear: registers 8s..8s+7 of hidden tile m = q/2, s = q%2: unit 32m + 16s + 8(j>>2) + 4h + (j&3)
            const int k = first ? 32 * (q >> 1) + 16 * h + 8 * (q & 1) + j : 32 * (q >> 1) + 16 * (q & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
            v = R.scale * (first ? W1 : W2)[(size_t)R.row * KDIM + k];
          }
          tile[((size_t)q * 64 + lane) * 8 + j] = f32_to_bf16_rne(v);
        }
      }
    for (int r = 0; r < 32; ++r) {
      const TileRow& R = tr.rows[(size_t)T * 32 + r];
      bias[(size_t)T * 32 + r] = R.row < 0 ? 0.f : R.scale * (first ? b1 : b2)[R.row];
    }
  }
  {  // A operand of the 0e bias product: lane (o, h), fragment s holds b[tile 3 + i][row o] for the mids i = 16 s + 8 h + j
    const int fan0e = conv_shape(IN, OUT).fan0e;
    uint16_t* const bt = reinterpret_cast<uint16_t*>(bias + (size_t)(tr.ntiles + 1) * 32);
    for (int s3 = 0; s3 < 3; ++s3)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int o = lane & 31, i = 16 * s3 + 8 * (lane >> 5) + j;
          bt[((size_t)s3 * 64 + lane) * 8 + j] = f32_to_bf16_rne(i < fan0e ? bias[(size_t)(3 + i) * 32 + o] : 0.f);
        }
  }
  return out;
}

// bf16x3 stream (tp_conv.hip::OpsBf16x3): every weight as the exact sum of three bf16 planes; (ntiles + 1) tiles of
// [6 k-steps x 3 planes][64 lanes][8 bf16] (18 KB, same k order as the bf16 stream), then the fp32 bias table.
static std::vector<float> pack_conv_stream_bf16x3(int IN, int OUT, const float* W1, const float* b1, const float* W2, const float* b2) {
  const TileRows tr = conv_tile_rows(IN, OUT, true);      // the merged layout of tp_conv.hip (both operand policies of that kernel)
  constexpr int TILE_BF16 = 3 * 32 * KDIM;   // 9216 bf16 = 18 KB
  std::vector<float> out(((size_t)(tr.ntiles + 1) * TILE_BF16 * 2 + (size_t)tr.ntiles * 32 * 4) / 4, 0.f);
  uint16_t* const w = reinterpret_cast<uint16_t*>(out.data());
  float* const bias_tab = reinterpret_cast<float*>(w + (size_t)(tr.ntiles + 1) * TILE_BF16);
  auto bf2f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
  for (int T = 0; T < tr.ntiles; ++T) {
    uint16_t* tile = w + (size_t)T * TILE_BF16;
    const bool first = T < 3;
    for (int q = 0; q < KDIM / 16; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        const TileRow& R = tr.rows[(size_t)T * 32 + r];
        for (int j = 0; j < 8; ++j) {
          const int k = first ? 32 * (q >> 1) + 16 * h + 8 * (q & 1) + j : 32 * (q >> 1) + 16 * (q & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
          const float v = R.row < 0 ? 0.f : R.scale * (first ? W1 : W2)[(size_t)R.row * KDIM + k];
          const uint16_t hi = f32_to_bf16_rne(v);
          const float r1 = v - bf2f(hi);
          const uint16_t mid = f32_to_bf16_rne(r1);
          const uint16_t lo = f32_to_bf16_rne(r1 - bf2f(mid));
          tile[((size_t)(3 * q + 0) * 64 + lane) * 8 + j] = hi;
          tile[((size_t)(3 * q + 1) * 64 + lane) * 8 + j] = mid;
          tile[((size_t)(3 * q + 2) * 64 + lane) * 8 + j] = lo;
        }
      }
    for (int r = 0; r < 32; ++r) {
      const TileRow& R = tr.rows[(size_t)T * 32 + r];
      bias_tab[(size_t)T * 32 + r] = R.row < 0 ? 0.f : R.scale * (first ? b1 : b2)[R.row];
    }
  }
  return out;
}

static int out_level_dim(int level) { return conv_shape(0, level).out_dim; }
static int in_level_dim(int level) { return conv_shape(level, 3).in_dim; }

static int build_conv_layer(cbd_engine* e, const std::string& prefix, int IN, int OUT, int groups, ConvLayerDev* L) {
  const ConvShape S = conv_shape(IN, OUT);
  L->in_level = IN; L->out_level = OUT; L->n_groups = groups;
  for (int g = 0; g < groups; ++g) {
    const std::string fc = groups == 1 ? prefix + ".fc" : prefix + ".fc." + std::to_string(g);
    const HostTensor *w0, *b0, *w1, *b1;
    CHK(need(e, fc + ".0.weight", {KDIM, KDIM}, &w0));
    CHK(need(e, fc + ".0.bias", {KDIM}, &b0));
    CHK(need(e, fc + ".3.weight", {S.weight_numel, KDIM}, &w1));
    CHK(need(e, fc + ".3.bias", {S.weight_numel}, &b1));
    std::vector<float> st = pack_conv_stream(IN, OUT, w0->data.data(), b0->data.data(), w1->data.data(), b1->data.data(), true);
    HIPCHK(e->wpool.upload(&L->wstream[g], st));
    HIPCHK(e->wpool.upload(&L->wstream_bf16[g], pack_conv_stream_bf16(IN, OUT, w0->data.data(), b0->data.data(), w1->data.data(), b1->data.data())));
    HIPCHK(e->wpool.upload(&L->wstream_x3[g], pack_conv_stream_bf16x3(IN, OUT, w0->data.data(), b0->data.data(), w1->data.data(), b1->data.data())));
    std::vector<float> sd((size_t)2 * NS * KDIM);   // node parts of the first Linear, k-major for node_proj_kernel
    for (int part = 0; part < 2; ++part)
      for (int k = 0; k < NS; ++k)
        for (int c = 0; c < KDIM; ++c) sd[((size_t)part * NS + k) * KDIM + c] = w0->data[(size_t)c * KDIM + NS * (1 + part) + k];
    HIPCHK(e->wpool.upload(&L->w1sd[g], sd));
  }
  // e3nn BatchNorm (eval) per output column
  const int nf = NS + NV + (OUT >= 2 ? NV : 0) + (OUT >= 3 ? NV : 0);
  const HostTensor *bw, *bb, *bm, *bv;
  CHK(need(e, prefix + ".batch_norm.weight", {nf}, &bw));
  CHK(need(e, prefix + ".batch_norm.bias", {NS}, &bb));
  CHK(need(e, prefix + ".batch_norm.running_mean", {NS}, &bm));
  CHK(need(e, prefix + ".batch_norm.running_var", {nf}, &bv));
  std::vector<float> sc(NODE_STRIDE, 0.f), mean(NODE_STRIDE, 0.f), bias(NODE_STRIDE, 0.f);
  for (int c = 0; c < S.out_dim; ++c) {
    int chn;
    if (c < COL_1O) chn = c;
    else if (c < COL_1E) chn = NS + (c - COL_1O) / 3;
    else if (c < COL_0O) chn = NS + NV + (c - COL_1E) / 3;
    else chn = NS + 2 * NV + (c - COL_0O);
    sc[c] = bw->data[chn] * (1.0f / std::sqrt(bv->data[chn] + 1e-5f));
    if (c < NS) { mean[c] = bm->data[c]; bias[c] = bb->data[c]; }
  }
  HIPCHK(e->wpool.upload(&L->bn_scale, sc));
  HIPCHK(e->wpool.upload(&L->bn_mean, mean));
  HIPCHK(e->wpool.upload(&L->bn_bias, bias));
  return 0;
}

static std::vector<float> transpose_block(const HostTensor* w, int col0, int ncol) {
  // w [out=32][in] -> [ncol][32] with element (k, o) = w[o][col0 + k]
  const int in = (int)w->shape[1], out = (int)w->shape[0];
  std::vector<float> t((size_t)ncol * out);
  for (int k = 0; k < ncol; ++k)
    for (int o = 0; o < out; ++o) t[(size_t)k * out + o] = w->data[(size_t)o * in + col0 + k];
  return t;
}

static int build_edge_mlp(cbd_engine* e, const std::string& prefix, int in_dim, int gauss_col0, int bond_col0,
                          const std::string& offset_key, MlpDev* M) {
  const HostTensor *w0, *b0, *w1, *b1, *off;
  CHK(need(e, prefix + ".0.weight", {NS, in_dim}, &w0));
  CHK(need(e, prefix + ".0.bias", {NS}, &b0));
  CHK(need(e, prefix + ".3.weight", {NS, NS}, &w1));
  CHK(need(e, prefix + ".3.bias", {NS}, &b1));
  CHK(need(e, offset_key, {32}, &off));
  HIPCHK(e->wpool.upload(&M->WgT, transpose_block(w0, gauss_col0, 32)));
  if (bond_col0 >= 0) HIPCHK(e->wpool.upload(&M->WbT, transpose_block(w0, bond_col0, 4)));
  HIPCHK(e->wpool.upload(&M->W1T, transpose_block(w1, 0, 32)));
  HIPCHK(e->wpool.upload(&M->b0, b0->data));
  HIPCHK(e->wpool.upload(&M->b1, b1->data));
  HIPCHK(e->wpool.upload(&M->offset, off->data));
  const double step = (double)off->data[1] - (double)off->data[0];   // GaussianSmearing.coeff (python float), score_model.py:672
  M->coeff = (float)(-0.5 / (step * step));
  return 0;
}

static int upload_named(cbd_engine* e, const std::string& k, std::initializer_list<int64_t> shape, const float** dev) {
  const HostTensor* t;
  CHK(need(e, k, shape, &t));
  float* p;
  HIPCHK(e->wpool.upload(&p, t->data));
  *dev = p;
  return 0;
}

// ======================================================================================================== C ABI
extern "C" {

const char* cbd_last_error(void) { return g_err.c_str(); }
const char* cbd_version(void) { return "cbdock-mi355x 0.1 (gfx950, fp32 MFMA)"; }

// The set-up stream of "async_setup".  HIP multiplexes the streams of a process onto a few hardware queues (four by default), and a
// stream that shares the queue of the stream a step-loop graph is running on waits for that graph: a set-up stream per engine lost
// its overlap for one engine in four (uploads blocked for the 470 ms the running wave still had to go).  Priority streams get queues of
// their own, but their mere existence cost the C4 bf16 leg of bench.py 10 % (226 -> 197 poses/s, either bf16 kernel; measured with one
// per engine and with one per process).  So: a small pool of ordinary streams per device, created back to back (they land on different
// queues), and every set-up PROBES for one whose queue is free -- an event recorded on the candidate must complete within a fraction
// of a millisecond.  Never destroyed.
struct SetupPool {
  static constexpr int N = 4;
  hipStream_t s[N] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev[N] = {nullptr, nullptr, nullptr, nullptr};
};
static std::mutex g_setup_mu;
static std::map<int, SetupPool> g_setup_pools;

static int pick_setup_stream(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> lock(g_setup_mu);
  auto it = g_setup_pools.find(device);
  if (it == g_setup_pools.end()) {
    SetupPool p;
    for (int k = 0; k < SetupPool::N; ++k) {
      HIPCHK(hipStreamCreateWithFlags(&p.s[k], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&p.ev[k], hipEventDisableTiming));
    }
    it = g_setup_pools.emplace(device, p).first;
  }
  SetupPool& p = it->second;
  for (int k = 0; k < SetupPool::N; ++k) {
    HIPCHK(hipEventRecord(p.ev[k], p.s[k]));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipEventQuery(p.ev[k]);
      if (q == hipSuccess) { *out = p.s[k]; return 0; }
      if (q != hipErrorNotReady) HIPCHK(q);
      (void)hipGetLastError();      // "not ready" is an answer, not an error: it must not surface at a later hipGetLastError() of this thread
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 400.0) break;
    }
  }
  *out = p.s[0];      // every queue is busy: wait on the first one
  return 0;
}

int cbd_create(const cbd_config* cfg, cbd_engine** out) {
  if (!cfg || !out) return fail(CBD_ERR_ARG, "null argument");
  if (cfg->ns != NS || cfg->nv != NV || cfg->num_conv_layers != 5 || cfg->num_prot_emb_layers != 3)
    return fail(CBD_ERR_ARG, "unsupported architecture: the engine implements ns=32, nv=6, 3 embedding + 5 interaction layers");
  if (cfg->lm_embedding_dim != 0 && cfg->lm_embedding_dim != 1280) return fail(CBD_ERR_ARG, "lm_embedding_dim must be 0 or 1280");
  if (cfg->max_batch <= 0) return fail(CBD_ERR_ARG, "max_batch must be positive");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(CBD_ERR_ARG, "device %d not available (%d devices)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  cbd_engine* e = new cbd_engine();
  e->cfg = *cfg;
  if (const char* p = getenv("CBD_PRECISION")) e->use_bf16 = std::max(0, std::min(2, atoi(p)));   // test hook: default operand policy
#ifdef CBD_EXPERIMENTS
  if (const char* p = getenv("CBD_BF16_ROLES")) e->bf16_roles = std::max(0, std::min(2, atoi(p)));                     // test hook: role split of the bf16 policy
#endif
  if (const char* p = getenv("CBD_BF16_STATIONARY")) e->bf16_stat = atoi(p) != 0;                                       // test hook: register-stationary bf16 kernel
  HIPCHK(hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, cfg->device));
#ifdef CBD_EXPERIMENTS
  if (const char* p = getenv("CBD_BF16P_WGS")) e->n_cus = std::max(1, atoi(p));                   // diagnostic: workgroups of the persistent kernel
#endif
  HIPCHK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&e->own, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&e->ev_last, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->ev_a, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->ev_b, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->desc_dev), sizeof(PoseBatch)));
  *out = e;
  return 0;
}

int cbd_destroy(cbd_engine* e) {
  if (!e) return 0;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  drop_graphs(e);
  if (e->desc_dev) (void)hipFree(e->desc_dev);
  e->wpool.release(); e->cpool.release(); e->bpool.release();
  for (auto& p : e->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (auto& p : e->gev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->own) (void)hipStreamDestroy(e->own);
  if (e->ev_last) (void)hipEventDestroy(e->ev_last);
  if (e->ev_a) (void)hipEventDestroy(e->ev_a);
  if (e->ev_b) (void)hipEventDestroy(e->ev_b);
  delete e;
  return 0;
}

int cbd_load_weight(cbd_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  if (!e || !name || (!data && ndim > 0 && shape[0] != 0)) return fail(CBD_ERR_ARG, "null argument");
  const std::string k(name);
  for (const char* pre : {"final_conv.tp.", "tor_bond_conv.tp.", "final_tp_tor."})
    if (k.rfind(pre, 0) == 0) return 0;   // e3nn persistent buffers: arithmetic is hard-wired
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  e->host_w[k] = std::move(t);
  e->weights_ready = false;
  return 0;
}

int cbd_finalize_weights(cbd_engine* e) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->wpool.release();
  const int lm = e->cfg.lm_embedding_dim;
  for (int l = 0; l < 3; ++l) {
    CHK(build_conv_layer(e, "rec_emb_layers." + std::to_string(l), l, l + 1, 1, &e->rec_emb[l]));
    CHK(build_conv_layer(e, "lig_emb_layers." + std::to_string(l), l, l + 1, 1, &e->lig_emb[l]));
  }
  for (int l = 0; l < 5; ++l) CHK(build_conv_layer(e, "conv_layers." + std::to_string(l), 3, 3, l == 4 ? 2 : 4, &e->conv[l]));
  // edge MLPs.  Input column order: lig [bond4 | sigma | gauss], cross [sigma | gauss], rec [gauss], center [gauss | sigma]
  CHK(build_edge_mlp(e, "lig_edge_embedding", 68, 36, 0, "lig_distance_expansion.offset", &e->m_lig_edge));
  CHK(build_edge_mlp(e, "cross_edge_embedding", 64, 32, -1, "cross_distance_expansion.offset", &e->m_cross));
  CHK(build_edge_mlp(e, "rec_edge_embedding", 32, 0, -1, "rec_distance_expansion.offset", &e->m_rec_edge));
  CHK(build_edge_mlp(e, "center_edge_embedding", 64, 0, -1, "center_distance_expansion.offset", &e->m_center));
  if (!e->cfg.no_torsion) CHK(build_edge_mlp(e, "final_edge_embedding", 32, 0, -1, "lig_distance_expansion.offset", &e->m_final_edge));
  // per-step small weights
  StepWeights& sw = e->sw;
  CHK(upload_named(e, "rec_sigma_embedding.0.weight", {32, 32}, &sw.rec_sig_w0));
  CHK(upload_named(e, "rec_sigma_embedding.0.bias", {32}, &sw.rec_sig_b0));
  CHK(upload_named(e, "rec_sigma_embedding.3.weight", {32, 32}, &sw.rec_sig_w1));
  CHK(upload_named(e, "rec_sigma_embedding.3.bias", {32}, &sw.rec_sig_b1));
  CHK(upload_named(e, "lig_edge_embedding.0.weight", {32, 68}, &sw.lig_edge_w0));
  CHK(upload_named(e, "lig_edge_embedding.0.bias", {32}, &sw.lig_edge_b0));
  CHK(upload_named(e, "cross_edge_embedding.0.weight", {32, 64}, &sw.cross_w0));
  CHK(upload_named(e, "cross_edge_embedding.0.bias", {32}, &sw.cross_b0));
  CHK(upload_named(e, "center_edge_embedding.0.weight", {32, 64}, &sw.center_w0));
  CHK(upload_named(e, "center_edge_embedding.0.bias", {32}, &sw.center_b0));
  CHK(upload_named(e, "lig_node_embedding.additional_features_embedder.weight", {32, 64}, &sw.lig_node_w));
  CHK(upload_named(e, "lig_node_embedding.additional_features_embedder.bias", {32}, &sw.lig_node_b));
  CHK(upload_named(e, "tr_final_layer.0.weight", {32, 33}, &sw.tr_w0));
  CHK(upload_named(e, "tr_final_layer.0.bias", {32}, &sw.tr_b0));
  CHK(upload_named(e, "rot_final_layer.0.weight", {32, 33}, &sw.rot_w0));
  CHK(upload_named(e, "rot_final_layer.0.bias", {32}, &sw.rot_b0));
  // ligand atom embedding tables stay on the host (folded into lig_static32 per complex)
  static const int dims[16] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};   // datasets/process_mols.py:95-112
  e->lig_emb_tables.clear();
  for (int i = 0; i < 16; ++i) {
    const HostTensor* t;
    CHK(need(e, "lig_node_embedding.atom_embedding_list." + std::to_string(i) + ".weight", {dims[i], 32}, &t));
    e->lig_emb_tables.push_back(t->data);
  }
  e->lig_node_w_host = find_w(e, "lig_node_embedding.additional_features_embedder.weight")->data;
  // receptor node encoder
  const float* tmp;
  CHK(upload_named(e, "rec_node_embedding.atom_embedding_list.0.weight", {38, 32}, &tmp));
  e->rec_emb_table = const_cast<float*>(tmp);
  if (lm > 0) {
    CHK(upload_named(e, "rec_node_embedding.additional_features_embedder.weight", {32, 32 + lm}, &tmp));
    e->rec_node_w = const_cast<float*>(tmp);
    CHK(upload_named(e, "rec_node_embedding.additional_features_embedder.bias", {32}, &tmp));
    e->rec_node_b = const_cast<float*>(tmp);
  }
  // centre head
  CenterHead& ch = e->ch;
  ch.ce_WgT = e->m_center.WgT; ch.ce_W1T = e->m_center.W1T; ch.ce_b1 = e->m_center.b1; ch.offset = e->m_center.offset;
  ch.coeff = e->m_center.coeff;
  CHK(upload_named(e, "final_conv.fc.0.weight", {64, 64}, &ch.fc_w0));
  CHK(upload_named(e, "final_conv.fc.0.bias", {64}, &ch.fc_b0));
  CHK(upload_named(e, "final_conv.fc.3.weight", {124, 64}, &ch.fc_w1));
  CHK(upload_named(e, "final_conv.fc.3.bias", {124}, &ch.fc_b1));
  {
    const HostTensor *bw, *bv;
    CHK(need(e, "final_conv.batch_norm.weight", {4}, &bw));
    CHK(need(e, "final_conv.batch_norm.running_var", {4}, &bv));
    std::vector<float> sc(4);
    for (int c = 0; c < 4; ++c) sc[c] = bw->data[c] * (1.0f / std::sqrt(bv->data[c] + 1e-5f));
    float* p;
    HIPCHK(e->wpool.upload(&p, sc));
    ch.bn_scale = p;
    for (int which = 0; which < 2; ++which) {
      const std::string pre = which ? "rot_final_layer" : "tr_final_layer";
      const HostTensor *w0, *w1, *b1;
      CHK(need(e, pre + ".0.weight", {32, 33}, &w0));
      CHK(need(e, pre + ".3.weight", {1, 32}, &w1));
      CHK(need(e, pre + ".3.bias", {1}, &b1));
      std::vector<float> col(32);
      for (int o = 0; o < 32; ++o) col[o] = w0->data[(size_t)o * 33];
      float *pc, *pw, *pb;
      HIPCHK(e->wpool.upload(&pc, col));
      HIPCHK(e->wpool.upload(&pw, w1->data));
      HIPCHK(e->wpool.upload(&pb, b1->data));
      if (which) { ch.rot_w0n = pc; ch.rot_w1 = pw; ch.rot_b1 = pb; } else { ch.tr_w0n = pc; ch.tr_w1 = pw; ch.tr_b1 = pb; }
    }
  }
  // torsion head
  if (!e->cfg.no_torsion) {
    BondHead& bh = e->bh;
    bh.fe.part = e->m_final_edge.b0; bh.fe.WgT = e->m_final_edge.WgT; bh.fe.WbT = nullptr; bh.fe.W1T = e->m_final_edge.W1T;
    bh.fe.b1 = e->m_final_edge.b1; bh.fe.offset = e->m_final_edge.offset; bh.fe.coeff = e->m_final_edge.coeff;
    CHK(upload_named(e, "tor_bond_conv.fc.0.weight", {96, 96}, &bh.fc_w0));
    CHK(upload_named(e, "tor_bond_conv.fc.0.bias", {96}, &bh.fc_b0));
    CHK(upload_named(e, "tor_bond_conv.fc.3.weight", {384, 96}, &bh.fc_w1));
    CHK(upload_named(e, "tor_bond_conv.fc.3.bias", {384}, &bh.fc_b1));
    {
      const HostTensor *w0, *b0, *w1, *b1;
      CHK(need(e, "tor_bond_conv.fc.0.weight", {96, 96}, &w0)); CHK(need(e, "tor_bond_conv.fc.0.bias", {96}, &b0));
      CHK(need(e, "tor_bond_conv.fc.3.weight", {384, 96}, &w1)); CHK(need(e, "tor_bond_conv.fc.3.bias", {384}, &b1));
      HIPCHK(e->wpool.upload(&e->bond_stream, pack_rows_f32(bond_tile_rows(), w0->data.data(), b0->data.data(), w1->data.data(), b1->data.data())));
    }
    const HostTensor *bw, *bb, *bm, *bv;
    CHK(need(e, "tor_bond_conv.batch_norm.weight", {64}, &bw));
    CHK(need(e, "tor_bond_conv.batch_norm.bias", {32}, &bb));
    CHK(need(e, "tor_bond_conv.batch_norm.running_mean", {32}, &bm));
    CHK(need(e, "tor_bond_conv.batch_norm.running_var", {64}, &bv));
    // irreps 32x0o + 32x0e: columns 0..31 pseudoscalars (scale only), 32..63 scalars (mean/bias index c-32)
    std::vector<float> sc(64), mean(64, 0.f), bias(64, 0.f);
    for (int c = 0; c < 64; ++c) {
      sc[c] = bw->data[c] * (1.0f / std::sqrt(bv->data[c] + 1e-5f));
      if (c >= 32) { mean[c] = bm->data[c - 32]; bias[c] = bb->data[c - 32]; }
    }
    float *p0, *p1, *p2;
    HIPCHK(e->wpool.upload(&p0, sc)); HIPCHK(e->wpool.upload(&p1, mean)); HIPCHK(e->wpool.upload(&p2, bias));
    bh.bn_scale = p0; bh.bn_mean = p1; bh.bn_bias = p2;
    CHK(upload_named(e, "tor_final_layer.0.weight", {32, 64}, &bh.tf_w0));
    CHK(upload_named(e, "tor_final_layer.3.weight", {1, 32}, &bh.tf_w1));
  }
  HIPCHK(hipDeviceSynchronize());
  e->weights_ready = true;
  e->complex_ready = false;   // the receptor embedding depends on the weights
  return 0;
}

// ======================================================================================================== complex
static EdgeMlp make_mlp(const MlpDev& m, const float* part) {
  EdgeMlp r{};
  r.part = part ? part : m.b0; r.WgT = m.WgT; r.WbT = m.WbT; r.W1T = m.W1T; r.b1 = m.b1; r.offset = m.offset; r.coeff = m.coeff;
  return r;
}


// Records `ev` on `s`.  Under stream capture the record becomes an explicit event-record node appended to the captured graph (the node
// is re-executed by every replay and the event can be queried from the host afterwards; an event recorded through hipEventRecord on a
// capturing stream belongs to the capture and cannot -- "invalid resource handle", measured in round 2; hipEventRecordWithFlags with
// hipEventRecordExternal is refused with "invalid argument" by this runtime).
static int record_event(hipEvent_t ev, hipStream_t s, bool capturing) {
  if (!capturing) { HIPCHK(hipEventRecord(ev, s)); return 0; }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t n_deps = 0;
  HIPCHK(hipStreamGetCaptureInfo_v2(s, &cs, &id, &graph, &deps, &n_deps));
  hipGraphNode_t node = nullptr;
  HIPCHK(hipGraphAddEventRecordNode(&node, graph, deps, n_deps, ev));
  HIPCHK(hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies));
  return 0;
}

static int launch_conv_timed(cbd_engine* e, const ConvLayerDev& L, const ConvArgs& a, int grid, hipStream_t s, const ConvArgs* resident = nullptr) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool cap = false;
  if (e->timing) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    HIPCHK(hipStreamIsCapturing(s, &cs));
    cap = cs == hipStreamCaptureStatusActive;
    cbd_engine::EvPool& pool = cap ? e->gev_pool : e->ev_pool;
    size_t& used = cap ? e->gev_used : e->ev_used;
    if (used == pool.size()) {
      hipEvent_t a0, a1;
      HIPCHK(hipEventCreate(&a0)); HIPCHK(hipEventCreate(&a1));
      pool.push_back({a0, a1});
    }
    e0 = pool[used].first; e1 = pool[used].second;
    ++used;
    CHK(record_event(e0, s, cap));
  }
  if (e->use_bf16 == 1 && e->bf16_stat && e->bf16_roles == 0 && L.in_level == 3 && L.out_level == 3 && tp_conv_bf16s_fits(a)) HIPCHK(launch_tp_conv_bf16s(a, e->n_cus, s));
  else if (e->use_bf16 == 1) HIPCHK(launch_tp_conv_bf16(L.in_level, L.out_level, a, grid, s));
  else if (e->use_bf16 == 2) HIPCHK(launch_tp_conv_x3(L.in_level, L.out_level, a, grid, s));
  else HIPCHK(launch_tp_conv(L.in_level, L.out_level, a, grid, s));
#ifdef CBD_EXPERIMENTS
  if (resident) HIPCHK(launch_tp_conv_bf16p(*resident, e->n_cus, s));
#else
  (void)resident;
#endif
  if (e->timing) CHK(record_event(e1, s, cap));
  return 0;
}

// The edge groups ONE pose batch contributes to a tensor-product launch.
struct ConvJob {
  cbd_engine* e = nullptr;
  ConvGroupH g[10];                // 4 edge groups, or ll + three slices of each of the other three (bf16 role split)
  int caps[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int widx[10] = {0, 1, 2, 3, 0, 0, 0, 0, 0, 0};      // which FCBlock of the layer each group uses
  int n_groups = 0;
  const float* node_in = nullptr;
};

// One tensor-product launch (+ the node-projection launch in front of it) over the groups of all jobs: the batches of up to eight
// complexes share every launch of the step loop, so a launch carries several times the waves of a single 40-pose batch (the
// per-launch drain of the long-lived waves is amortised).  Timed by jobs[0].e.
static int run_conv(const ConvLayerDev& L, const ConvJob* jobs, int n_jobs, hipStream_t s, bool side) {
  cbd_engine* e0 = jobs[0].e;
  constexpr int MAX_ALL = 8 * 10;      // up to eight co-scheduled batches of ten slices
  static thread_local ConvGroup all[MAX_ALL];
  int caps[MAX_ALL];
  int n_all = 0;
  ProjArgs pa{};
  const ConvShape S = conv_shape(L.in_level, L.out_level);
  for (int q = 0; q < n_jobs; ++q) {
    const ConvJob& J = jobs[q];
    cbd_engine* e = J.e;
    if (n_all + J.n_groups > MAX_ALL) return fail(CBD_ERR_STATE, "too many edge groups in one launch");
    const int base = side ? 8 : 0;
    int slot_of[4] = {-1, -1, -1, -1}, n_slots = 0;     // indexed by FCBlock (widx)
    for (int g = 0; g < J.n_groups; ++g) {
      ConvGroup& G = all[n_all];
      caps[n_all++] = J.caps[g];
      G = J.g[g];
      const int w = J.widx[g];
      G.wstream = (e0->use_bf16 == 1 ? L.wstream_bf16 : e0->use_bf16 == 2 ? L.wstream_x3 : L.wstream)[w];
      G.node_in = J.node_in;
      if (G.i0e_hi == 0 && G.vec_on == 0) {   // not a virtual slice: the whole weight-tile chain
        G.i0e_lo = 0; G.i0e_hi = S.t0e; G.vec_on = 1;
      }
      if (e0->use_bf16 != 1) {
        // per-node projections of the first Linear's node parts, one job per distinct (FCBlock, role); virtual slices share them.
        // (The plain-bf16 policy keeps the whole first Linear in the edge kernel.)
        if (slot_of[w] < 0) {
          slot_of[w] = n_slots++;
          if (base + 2 * slot_of[w] + 1 >= (int)(sizeof(e->proj) / sizeof(e->proj[0]))) return fail(CBD_ERR_STATE, "projection slots exhausted");
          if (pa.n_jobs + 2 > PROJ_MAX_JOBS) return fail(CBD_ERR_STATE, "projection jobs exhausted");
          float* ps = e->proj[base + 2 * slot_of[w]];
          float* pd = e->proj[base + 2 * slot_of[w] + 1];
          pa.job[pa.n_jobs++] = ProjJob{L.w1sd[w], ps, J.node_in, J.g[g].src_lo, J.g[g].src_n};
          pa.job[pa.n_jobs++] = ProjJob{L.w1sd[w] + (size_t)NS * KDIM, pd, J.node_in, J.g[g].dst_lo, J.g[g].dst_n};
        }
        G.psrc = e->proj[base + 2 * slot_of[w]];
        G.pdst = e->proj[base + 2 * slot_of[w] + 1];
      }
    }
  }
  if (pa.n_jobs) HIPCHK(launch_node_proj(pa, s));
  // role split with resident weights (bf16_roles == 2): the 0e-only slices go to the persistent kernel (tp_conv_bf16p.hip), everything
  // else stays with the streaming one
  const bool resident = e0->use_bf16 == 1 && e0->bf16_roles == 2 && L.in_level == 3 && L.out_level == 3;
  const int edges_per_wg = e0->use_bf16 == 1 ? 64 : CONV_WG_EDGES;                    // bf16: one wave per 64 edges
  ConvArgs a{}, ap{};
  a.stamps = e0->stamps_dev;
  int grid = 0;
  for (int i = 0; i < n_all; ++i) {
    const ConvGroup& G = all[i];
    const bool to_resident = resident && G.vec_on == 0 && G.i0e_hi > G.i0e_lo;
    ConvArgs& t = to_resident ? ap : a;
    if (t.n_groups == CONV_MAX_GROUPS) return fail(CBD_ERR_STATE, "too many edge groups in one launch");
    t.g[t.n_groups++] = G;
    if (!to_resident) grid += (caps[i] + edges_per_wg - 1) / edges_per_wg;
  }
  if (ap.n_groups) { ap.stamps = a.stamps; a.stamps = nullptr; }      // diagnostic stamps: the persistent kernel's
  return launch_conv_timed(e0, L, a, grid, s, ap.n_groups ? &ap : nullptr);
}

static FinGroup fin_group(const ConvGroup& g, const int* start, const int* cnt, int node_mod = 0) {
  FinGroup f{};
  f.start = start; f.cnt = cnt; f.total = g.count; f.first_sum = g.first_sum; f.last_sum = g.last_sum; f.run_acc = g.run_acc;
  f.node_mod = node_mod;
  f.deg_weight = 1;
  return f;
}

static int run_finalize(cbd_engine* e, const ConvLayerDev& L, const float* node_in, float* node_out, const FinGroup* groups,
                        int n_groups, int n_nodes, int node_off, hipStream_t s) {
  FinArgs fa{};
  fa.n_groups = n_groups;
  for (int g = 0; g < n_groups; ++g) fa.g[g] = groups[g];
  HIPCHK(launch_conv_finalize(fa, node_in, node_out, L.bn_scale, L.bn_mean, L.bn_bias, n_nodes, in_level_dim(L.in_level),
                              out_level_dim(L.out_level), node_off, s));
  return 0;
}

// Time-independent receptor embedding for ONE copy of the receptor (score_model.py:297-320): node encoder,
// edge embedding, three rec_emb_layers; result cached in rec_static (the reference caches it on the batch object).
static int embed_receptor(cbd_engine* e, hipStream_t s) {
  const GraphStatic& gs = e->gs;
  const int Nr = gs.Nr, Err = gs.Err, lm = e->cfg.lm_embedding_dim;
  HIPCHK(launch_rec_node_embed(e->d_rec_x, Nr, lm, e->rec_emb_table, e->rec_node_w, e->rec_node_b, e->X0, s));
  HIPCHK(launch_edge_geom(gs.rec_pos, e->d_src0, e->d_dst0, Err, e->d_vec0, e->d_dist0, s));
  {
    EdgeMlpArgs ma{};
    ma.n = 1;
    ma.seg[0] = EdgeSeg{make_mlp(e->m_rec_edge, nullptr), e->d_dist0, nullptr, nullptr, Err, e->rr_attr0};
    HIPCHK(launch_edge_mlp(ma, s));
  }
  HIPCHK(hipMemcpyAsync(e->rr_count_dev, &gs.Err, sizeof(int), hipMemcpyHostToDevice, s));
  float* in = e->X0;
  float* out = e->X1;
  for (int l = 0; l < 3; ++l) {
    ConvJob J;
    J.e = e; J.n_groups = 1; J.node_in = in; J.caps[0] = Err; J.widx[0] = 0;
    ConvGroupH& g = J.g[0];
    g.src = e->d_src0; g.dst = e->d_dst0; g.attr_idx = e->d_ident; g.vec = e->d_vec0; g.attr = e->rr_attr0; g.count = e->rr_count_dev;
    g.first_sum = e->fsum[4]; g.last_sum = e->lsum[4]; g.run_acc = e->racc[4];
    g.src_lo = 0; g.src_n = Nr; g.dst_lo = 0; g.dst_n = Nr;
    CHK(run_conv(e->rec_emb[l], &J, 1, s, false));
    const FinGroup fg = fin_group(g, e->rr0_start, e->d_deg0);
    CHK(run_finalize(e, e->rec_emb[l], in, out, &fg, 1, Nr, 0, s));
    std::swap(in, out);
  }
  HIPCHK(hipMemcpyAsync(e->rec_static, in, (size_t)Nr * NODE_STRIDE * 4, hipMemcpyDeviceToDevice, s));
  // X0/X1 rows [0, Nr) were used as scratch: the ligand rows of the next forward are rewritten in full, but clear the
  // columns a lower-level layer does not write
  HIPCHK(hipMemsetAsync(e->X0, 0, (size_t)Nr * NODE_STRIDE * 4, s));
  HIPCHK(hipMemsetAsync(e->X1, 0, (size_t)Nr * NODE_STRIDE * 4, s));
  return 0;
}

int cbd_set_complex(cbd_engine* e, int32_t Nl, int32_t Nr, int32_t nbd, int32_t R, int32_t Err, const int64_t* lig_x,
                    const int64_t* bond_index, const float* bond_attr, const uint8_t* edge_mask, const uint8_t* mask_rotate,
                    const float* rec_x, const float* rec_pos, const int64_t* rec_edge_index) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  if (!e->weights_ready) return fail(CBD_ERR_STATE, "cbd_finalize_weights must succeed before cbd_set_complex");
  if (Nl <= 0 || Nr <= 0 || nbd < 0 || R < 0 || Err < 0) return fail(CBD_ERR_ARG, "bad sizes");
  HIPCHK(hipSetDevice(e->cfg.device));
  const bool own_stream = e->async_setup && !e->sync_all;
  static const bool trace_setup = getenv("CBD_TRACE_SETUP") != nullptr;
  const auto t_in = std::chrono::steady_clock::now();
  auto since = [&](const char* what) {
    if (trace_setup) fprintf(stderr, "      [set_complex %p] %-22s +%.2f ms\n", (void*)e, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
  };
  if (own_stream) {      // only this engine's buffers are rewritten: wait for ITS last launches, leave the rest of the device alone
    if (e->last_used) HIPCHK(hipEventSynchronize(e->ev_last));
    CHK(pick_setup_stream(e->cfg.device, &e->setup));
  } else {
    HIPCHK(hipDeviceSynchronize());
  }
  since(own_stream ? "waited (own launches)" : "waited (device)");
  hipStream_t s = own_stream ? e->setup : nullptr;
  e->cpool.stream = e->bpool.stream = s;
  // the pools go back to the default stream on EVERY way out of this function (an early error return used to leave them
  // pointed at the set-up stream for whoever allocates from them next)
  struct PoolStreamGuard {
    cbd_engine* e;
    ~PoolStreamGuard() { e->cpool.stream = e->bpool.stream = nullptr; }
  } pool_stream_guard{e};
  e->last_used = false;
  e->sync_all = false;
  drop_graphs(e);
  e->g_pos = nullptr; e->g_S_cap = 0;
  e->cpool.reset(); e->bpool.reset();
  e->complex_ready = false;
  const int Bm = e->cfg.max_batch, lm = e->cfg.lm_embedding_dim;
  GraphStatic& gs = e->gs;
  gs = GraphStatic{};
  gs.Nl = Nl; gs.Nr = Nr; gs.R = R; gs.nbd = nbd; gs.Err = Err; gs.rec_off = Bm * Nl;

  // ---- ligand statics
  static const int dims[16] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};
  std::vector<float> lig_static((size_t)Nl * 32);
  for (int a = 0; a < Nl; ++a) {
    float emb[32] = {0};
    for (int f = 0; f < 16; ++f) {
      const int64_t v = lig_x[(size_t)a * 16 + f];
      if (v < 0 || v >= dims[f]) return fail(CBD_ERR_ARG, "ligand feature %d of atom %d out of range", f, a);
      for (int c = 0; c < 32; ++c) emb[c] += e->lig_emb_tables[f][(size_t)v * 32 + c];
    }
    for (int o = 0; o < 32; ++o) {   // first half of additional_features_embedder (the sigma half is per step)
      float acc = 0.f;
      for (int k = 0; k < 32; ++k) acc = std::fma(e->lig_node_w_host[(size_t)o * 64 + k], emb[k], acc);
      lig_static[(size_t)a * 32 + o] = acc;
    }
  }
  HIPCHK(e->cpool.upload(&e->lig_static32, lig_static));
  // bonds sorted by source atom
  std::vector<int> order(nbd);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bond_index[a] < bond_index[b]; });
  std::vector<int> bond_row(Nl + 1, 0), bond_dst(nbd);
  std::vector<float> battr((size_t)nbd * 4);
  int max_bdeg = 0;
  for (int k = 0; k < nbd; ++k) {
    const int o = order[k];
    const int64_t s = bond_index[o], d = bond_index[nbd + o];
    if (s < 0 || s >= Nl || d < 0 || d >= Nl) return fail(CBD_ERR_ARG, "bond index out of range");
    bond_row[s + 1]++;
    bond_dst[k] = (int)d;
    for (int c = 0; c < 4; ++c) battr[(size_t)k * 4 + c] = bond_attr[(size_t)o * 4 + c];
  }
  for (int a = 0; a < Nl; ++a) { max_bdeg = std::max(max_bdeg, bond_row[a + 1]); bond_row[a + 1] += bond_row[a]; }
  std::vector<int> rot_u, rot_v;
  for (int k = 0; k < nbd; ++k)
    if (edge_mask[k]) { rot_u.push_back((int)bond_index[k]); rot_v.push_back((int)bond_index[nbd + k]); }
  if ((int)rot_u.size() != R) return fail(CBD_ERR_ARG, "edge_mask selects %d bonds but R = %d", (int)rot_u.size(), R);
  for (int r = 0; r < R; ++r)
    if (mask_rotate[(size_t)r * Nl + rot_u[r]] || !mask_rotate[(size_t)r * Nl + rot_v[r]])
      return fail(CBD_ERR_ARG, "mask_rotate violates the u-outside / v-inside convention (utils/torsion.py:81-82)");
  int *d_bond_row, *d_bond_dst, *d_rot_u, *d_rot_v;
  float* d_battr;
  uint8_t* d_mask;
  HIPCHK(e->cpool.upload(&d_bond_row, bond_row)); HIPCHK(e->cpool.upload(&d_bond_dst, bond_dst));
  HIPCHK(e->cpool.upload(&d_battr, battr));
  HIPCHK(e->cpool.upload(&d_rot_u, rot_u)); HIPCHK(e->cpool.upload(&d_rot_v, rot_v));
  HIPCHK(e->cpool.upload(&d_mask, std::vector<uint8_t>(mask_rotate, mask_rotate + (size_t)R * Nl)));
  gs.bond_row = d_bond_row; gs.bond_dst = d_bond_dst; gs.bond_attr = d_battr; gs.rot_u = d_rot_u; gs.rot_v = d_rot_v;
  gs.mask_rotate = d_mask;
  // radius_graph keeps the first cap+1 hits INCLUDING self and then drops self: an atom whose own index comes after its
  // first cap+1 neighbours keeps cap+1 of them
  e->cap_ll_per_sample = nbd + Nl * std::min(Nl - 1, e->cfg.lig_radius_cap + 1);

  // ---- receptor statics: kNN edges sorted by aggregating node (row 0)
  std::vector<int> rorder(Err);
  std::iota(rorder.begin(), rorder.end(), 0);
  std::stable_sort(rorder.begin(), rorder.end(), [&](int a, int b) { return rec_edge_index[a] < rec_edge_index[b]; });
  std::vector<int> src0(Err), dst0(Err), deg0(Nr, 0), ident(Err);
  for (int k = 0; k < Err; ++k) {
    const int64_t s = rec_edge_index[rorder[k]], d = rec_edge_index[Err + rorder[k]];
    if (s < 0 || s >= Nr || d < 0 || d >= Nr) return fail(CBD_ERR_ARG, "receptor edge index out of range");
    src0[k] = (int)s; dst0[k] = (int)d; deg0[s]++; ident[k] = k;
  }
  float* d_rec_pos;
  int *d_src0, *d_dst0, *d_deg0, *d_ident;
  HIPCHK(e->cpool.upload(&d_rec_pos, std::vector<float>(rec_pos, rec_pos + (size_t)Nr * 3)));
  HIPCHK(e->cpool.upload(&d_src0, src0)); HIPCHK(e->cpool.upload(&d_dst0, dst0)); HIPCHK(e->cpool.upload(&d_deg0, deg0));
  HIPCHK(e->cpool.upload(&d_ident, ident));
  gs.rec_pos = d_rec_pos; gs.rr_deg0 = d_deg0;
  float *d_rec_x, *d_vec0, *d_dist0;
  HIPCHK(e->cpool.upload(&d_rec_x, std::vector<float>(rec_x, rec_x + (size_t)Nr * (1 + lm))));
  HIPCHK(e->cpool.alloc(&d_vec0, (size_t)Err * 4)); HIPCHK(e->cpool.alloc(&d_dist0, (size_t)Err));
  HIPCHK(e->cpool.alloc(&e->rr_attr0, (size_t)Err * 32)); HIPCHK(e->cpool.alloc(&e->rr_attr_t, (size_t)Err * 32));
  HIPCHK(e->cpool.alloc(&e->rec_static, (size_t)Nr * NODE_STRIDE));

  since("complex statics up");
  // ---- batch workspace (capacity max_batch)
  const int N = Bm * (Nl + Nr);
  e->n_nodes_cap = N;
  GraphDyn& gd = e->gd;
  gd = GraphDyn{};
  const size_t cap_ll = (size_t)Bm * e->cap_ll_per_sample, cap_x = (size_t)Bm * Nl * Nr;
  HIPCHK(e->bpool.alloc(&gd.cnt_ll, (size_t)Bm * Nl)); HIPCHK(e->bpool.alloc(&gd.cnt_lr, (size_t)Bm * Nl));
  HIPCHK(e->bpool.alloc(&gd.cnt_rl, (size_t)Bm * Nr));
  HIPCHK(e->bpool.alloc(&gd.start_ll, (size_t)Bm * Nl)); HIPCHK(e->bpool.alloc(&gd.start_lr, (size_t)Bm * Nl));
  HIPCHK(e->bpool.alloc(&gd.start_rl, (size_t)Bm * Nr));
  HIPCHK(e->bpool.alloc(&gd.counts, 8));
  HIPCHK(e->bpool.alloc(&gd.ll_src, cap_ll)); HIPCHK(e->bpool.alloc(&gd.ll_dst, cap_ll)); HIPCHK(e->bpool.alloc(&gd.ll_aidx, cap_ll));
  HIPCHK(e->bpool.alloc(&gd.ll_vec, cap_ll * 4)); HIPCHK(e->bpool.alloc(&gd.ll_dist, cap_ll)); HIPCHK(e->bpool.alloc(&gd.ll_bond4, cap_ll * 4));
  HIPCHK(e->bpool.alloc(&gd.lr_src, cap_x)); HIPCHK(e->bpool.alloc(&gd.lr_dst, cap_x)); HIPCHK(e->bpool.alloc(&gd.lr_aidx, cap_x));
  HIPCHK(e->bpool.alloc(&gd.lr_vec, cap_x * 4)); HIPCHK(e->bpool.alloc(&gd.lr_dist, cap_x));
  HIPCHK(e->bpool.alloc(&gd.rl_src, cap_x)); HIPCHK(e->bpool.alloc(&gd.rl_dst, cap_x)); HIPCHK(e->bpool.alloc(&gd.rl_aidx, cap_x));
  HIPCHK(e->bpool.alloc(&gd.rl_vec, cap_x * 4));
  HIPCHK(e->bpool.alloc(&gd.pair_eid, cap_x));
  HIPCHK(e->bpool.alloc(&e->ll_attr, cap_ll * 32)); HIPCHK(e->bpool.alloc(&e->lr_attr, cap_x * 32));
  HIPCHK(e->bpool.alloc(&e->X0, (size_t)N * NODE_STRIDE)); HIPCHK(e->bpool.alloc(&e->X1, (size_t)N * NODE_STRIDE));
  for (float*& pbuf : e->proj) HIPCHK(e->bpool.alloc(&pbuf, (size_t)N * KDIM));
  HIPCHK(hipMemsetAsync(e->X0, 0, (size_t)N * NODE_STRIDE * 4, s)); HIPCHK(hipMemsetAsync(e->X1, 0, (size_t)N * NODE_STRIDE * 4, s));
  float* vecs;
  HIPCHK(e->bpool.alloc(&vecs, 7 * 32));
  e->sv = StepVectors{vecs, vecs + 32, vecs + 64, vecs + 96, vecs + 128, vecs + 160, vecs + 192};
  e->sigma_cap = 64;
  HIPCHK(e->bpool.alloc(&e->sigma_emb_dev, (size_t)e->sigma_cap * 64));
  HIPCHK(e->bpool.alloc(&e->tr_out, (size_t)Bm * 3)); HIPCHK(e->bpool.alloc(&e->rot_out, (size_t)Bm * 3));
  HIPCHK(e->bpool.alloc(&e->tor_out, (size_t)Bm * std::max(R, 1)));
  HIPCHK(e->bpool.alloc(&e->center_msg, (size_t)Bm * Nl * 12));
  HIPCHK(e->bpool.alloc(&e->tor_nb, (size_t)Bm * std::max(R, 1) * 32)); HIPCHK(e->bpool.alloc(&e->tor_nb_cnt, (size_t)Bm * std::max(R, 1)));
  HIPCHK(e->bpool.alloc(&e->dbg_global, (size_t)Bm * 12)); HIPCHK(e->bpool.alloc(&e->dbg_torfeat, (size_t)Bm * std::max(R, 1) * 64));
  // batched receptor edges (independent of B: receptor rows start at rec_off)
  std::vector<int> bsrc((size_t)Bm * Err), bdst((size_t)Bm * Err), baidx((size_t)Bm * Err);
  for (int b = 0; b < Bm; ++b)
    for (int k = 0; k < Err; ++k) {
      bsrc[(size_t)b * Err + k] = gs.rec_off + b * Nr + src0[k];
      bdst[(size_t)b * Err + k] = gs.rec_off + b * Nr + dst0[k];
      baidx[(size_t)b * Err + k] = k;
    }
  HIPCHK(e->bpool.upload(&e->rr_src, bsrc)); HIPCHK(e->bpool.upload(&e->rr_dst, bdst)); HIPCHK(e->bpool.upload(&e->rr_aidx, baidx));
  HIPCHK(e->bpool.alloc(&e->rr_vec, (size_t)Bm * Err * 4));
  HIPCHK(e->bpool.alloc(&e->rr_count_dev, 1));
  {
    const size_t caps[5] = {cap_ll, cap_x, (size_t)Bm * Err, cap_x, (size_t)Err};
    for (int g = 0; g < 5; ++g) {
      const size_t tiles = (caps[g] + CONV_WG_EDGES - 1) / CONV_WG_EDGES + 1;
      const size_t sf = tiles * NODE_STRIDE, sr = (size_t)(g == 4 ? Nr : N) * NODE_STRIDE;
#ifdef CBD_EXPERIMENTS
      if (g >= 1 && g <= 3) {   // one block: [first | last | run_acc] of the group, then the same again for its second 0e slice
        const size_t D = (2 * sf + sr + 63) / 64 * 64;
        float* base = nullptr;
        HIPCHK(e->bpool.alloc(&base, 2 * D));
        e->fsum[g] = base; e->lsum[g] = base + sf; e->racc[g] = base + 2 * sf;
        e->piece_b_off[g] = (long long)D;
        continue;
      }
#endif
      HIPCHK(e->bpool.alloc(&e->fsum[g], sf));
      HIPCHK(e->bpool.alloc(&e->lsum[g], sf));
      HIPCHK(e->bpool.alloc(&e->racc[g], sr));
    }
    for (int k = 0; k < 2; ++k) {
      const size_t tiles = (cap_ll + CONV_WG_EDGES - 1) / CONV_WG_EDGES + 1;
      HIPCHK(e->bpool.alloc(&e->fsum_x[k], tiles * NODE_STRIDE));
      HIPCHK(e->bpool.alloc(&e->lsum_x[k], tiles * NODE_STRIDE));
      HIPCHK(e->bpool.alloc(&e->racc_x[k], (size_t)Bm * Nl * NODE_STRIDE));
    }
    std::vector<int> row0(Nr + 1, 0);
    for (int r = 0; r < Nr; ++r) row0[r + 1] = row0[r] + deg0[r];
    std::vector<int> bstart((size_t)Bm * Nr), bcnt((size_t)Bm * Nr);
    for (int b = 0; b < Bm; ++b)
      for (int r = 0; r < Nr; ++r) { bstart[(size_t)b * Nr + r] = b * Err + row0[r]; bcnt[(size_t)b * Nr + r] = deg0[r]; }
    HIPCHK(e->bpool.upload(&e->rr_start, bstart)); HIPCHK(e->bpool.upload(&e->rr_cnt, bcnt));
    row0.pop_back();
    HIPCHK(e->bpool.upload(&e->rr0_start, row0));
  }

  e->d_rec_x = d_rec_x; e->d_vec0 = d_vec0; e->d_dist0 = d_dist0; e->d_src0 = d_src0; e->d_dst0 = d_dst0;
  e->d_ident = d_ident; e->d_deg0 = d_deg0;
  HIPCHK(e->bpool.alloc(&e->stats_dev, 4));
  e->stamps_dev = nullptr;
#ifdef CBD_DIAG      // the stamping kernels exist in the diagnostic library only
  if ((getenv("CBD_CONV_VARIANT") && (atoi(getenv("CBD_CONV_VARIANT")) == 8 || atoi(getenv("CBD_CONV_VARIANT")) == 13)) ||
      (getenv("CBD_BF16_DIAG") && (atoi(getenv("CBD_BF16_DIAG")) >= 4 && atoi(getenv("CBD_BF16_DIAG")) <= 6))) {
    HIPCHK(e->bpool.alloc(&e->stamps_dev, 8192 * 8));
    HIPCHK(hipMemsetAsync(e->stamps_dev, 0, 8192 * 8 * 8, s));
  }
#endif
  HIPCHK(hipMemsetAsync(e->stats_dev, 0, 4 * sizeof(unsigned long long), s));
  fill_static_desc(e);
  since("workspace");
  CHK(embed_receptor(e, s));
  since("embed_receptor queued");
  for (int b = 0; b < Bm; ++b)
    HIPCHK(hipMemcpyAsync(e->rr_vec + (size_t)b * Err * 4, d_vec0, (size_t)Err * 16, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemsetAsync(e->X0, 0, (size_t)N * NODE_STRIDE * 4, s)); HIPCHK(hipMemsetAsync(e->X1, 0, (size_t)N * NODE_STRIDE * 4, s));
  HIPCHK(hipStreamSynchronize(s));
  since("done");
  e->complex_ready = true;
  return 0;
}

// ======================================================================================================== forward
static void snap(cbd_engine* e, const char* name, const float* dev, size_t n, hipStream_t s) {
  if (!e->keep_debug) return;
  std::vector<float> h(n);
  (void)hipStreamSynchronize(s);
  (void)hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost);
  e->dbg_snap.push_back({name, std::move(h)});
}

// The edge groups of one pose batch as the tensor-product kernel sees them: ligand-ligand, ligand->receptor, receptor->receptor,
// receptor->ligand, and the single-copy receptor graph whose layer-0 messages are shared by all samples.
struct BatchGroups { ConvGroupH ll, lr, rr, rl, rr_shared; };

static BatchGroups batch_groups(cbd_engine* e, int B) {
  const GraphStatic& gs = e->gs;
  const GraphDyn& gd = e->gd;
  const int nL = B * gs.Nl, nR = B * gs.Nr;
  BatchGroups G{};
  G.ll.src = gd.ll_src; G.ll.dst = gd.ll_dst; G.ll.attr_idx = gd.ll_aidx; G.ll.vec = gd.ll_vec; G.ll.attr = e->ll_attr; G.ll.count = gd.counts + 0;
  G.lr.src = gd.lr_src; G.lr.dst = gd.lr_dst; G.lr.attr_idx = gd.lr_aidx; G.lr.vec = gd.lr_vec; G.lr.attr = e->lr_attr; G.lr.count = gd.counts + 1;
  G.rr.src = e->rr_src; G.rr.dst = e->rr_dst; G.rr.attr_idx = e->rr_aidx; G.rr.vec = e->rr_vec; G.rr.attr = e->rr_attr_t; G.rr.count = gd.counts + 2;
  G.rl.src = gd.rl_src; G.rl.dst = gd.rl_dst; G.rl.attr_idx = gd.rl_aidx; G.rl.vec = gd.rl_vec; G.rl.attr = e->lr_attr; G.rl.count = gd.counts + 3;
  ConvGroupH* gg[4] = {&G.ll, &G.lr, &G.rr, &G.rl};
  for (int g = 0; g < 4; ++g) { gg[g]->first_sum = e->fsum[g]; gg[g]->last_sum = e->lsum[g]; gg[g]->run_acc = e->racc[g]; }
  // node-row ranges of the aggregating (src) and the read (dst) side: ligand rows [0, nL), receptor rows [rec_off, rec_off + nR)
  G.ll.src_lo = 0; G.ll.src_n = nL; G.ll.dst_lo = 0; G.ll.dst_n = nL;
  G.lr.src_lo = 0; G.lr.src_n = nL; G.lr.dst_lo = gs.rec_off; G.lr.dst_n = nR;
  G.rr.src_lo = gs.rec_off; G.rr.src_n = nR; G.rr.dst_lo = gs.rec_off; G.rr.dst_n = nR;
  G.rl.src_lo = gs.rec_off; G.rl.src_n = nR; G.rl.dst_lo = 0; G.rl.dst_n = nL;
  // cost of a 32-edge unit per role for the persistent bf16 kernel's work split, in 1/64 of a ligand->receptor unit (per-workgroup
  // lifetimes on C4, tools/conv_span_wg.py; ConvGroup::cost_w)
  G.ll.cost_w = 70; G.lr.cost_w = 64; G.rr.cost_w = 68; G.rl.cost_w = 66;      // (sweep: profiles/r06_n_split_weights.txt)
#ifdef CBD_DIAG      // diagnostic library only: CBD_S_WEIGHTS="ll,lr,rr,rl" (1/64) for tuning runs
  if (const char* p = getenv("CBD_S_WEIGHTS")) {
    int w[4] = {70, 64, 68, 66};
    if (sscanf(p, "%d,%d,%d,%d", &w[0], &w[1], &w[2], &w[3]) == 4) { G.ll.cost_w = w[0]; G.lr.cost_w = w[1]; G.rr.cost_w = w[2]; G.rl.cost_w = w[3]; }
  }
#endif
  // sample 0's receptor edges (the first Err entries of the batched arrays) with their own piece buffers
  G.rr_shared.src = e->rr_src; G.rr_shared.dst = e->rr_dst; G.rr_shared.attr_idx = e->rr_aidx; G.rr_shared.vec = e->rr_vec;
  G.rr_shared.attr = e->rr_attr_t; G.rr_shared.count = e->rr_count_dev;
  G.rr_shared.first_sum = e->fsum[4]; G.rr_shared.last_sum = e->lsum[4]; G.rr_shared.run_acc = e->racc[2];
  G.rr_shared.src_lo = gs.rec_off; G.rr_shared.src_n = gs.Nr; G.rr_shared.dst_lo = gs.rec_off; G.rr_shared.dst_n = gs.Nr;
  return G;
}

// The ligand embedding layers hold only ~10 edge tiles per pose (a fraction of one round of waves) and are bound by the length of a
// wave's weight-tile chain: the chain is split over two waves per edge tile (virtual slices of the ll group with their own pieces).
constexpr int EMB_SLICES = 2;
static void emb_slices(cbd_engine* e, const ConvGroupH& gll, int level, ConvGroupH (&sl)[EMB_SLICES]) {
  const ConvShape ES = conv_shape(level, level + 1);
  const int tvec = ES.t1o + ES.t1e + ES.t0o, nsl = EMB_SLICES;
  const int per = (ES.t0e + tvec + nsl - 1) / nsl;
  const int c_lo = ES.t0e - std::max(0, std::min(ES.t0e, per - tvec));   // 0e tiles the vector slice also takes
  for (int k = 0; k < nsl; ++k) {
    sl[k] = gll;
    if (k > 0) { sl[k].first_sum = e->fsum_x[k - 1]; sl[k].last_sum = e->lsum_x[k - 1]; sl[k].run_acc = e->racc_x[k - 1]; }
    if (k == nsl - 1) { sl[k].i0e_lo = c_lo; sl[k].i0e_hi = ES.t0e; sl[k].vec_on = 1; }
    else { sl[k].i0e_lo = c_lo * k / (nsl - 1); sl[k].i0e_hi = c_lo * (k + 1) / (nsl - 1); sl[k].vec_on = 0; }
  }
}

// Static part of the batch descriptor (everything that is fixed once the complex is set).
static void fill_static_desc(cbd_engine* e) {
  PoseBatch& D = e->desc_h;
  D = PoseBatch{};
  D.gs = e->gs;
  D.gd = e->gd;
  D.X[0] = e->X0; D.X[1] = e->X1;
  D.lig_static32 = e->lig_static32; D.rec_static = e->rec_static; D.rr_attr0 = e->rr_attr0; D.rr_attr_t = e->rr_attr_t;
  D.ll_attr = e->ll_attr; D.lr_attr = e->lr_attr;
  D.center_msg = e->center_msg; D.dbg_global = e->dbg_global; D.dbg_torfeat = e->dbg_torfeat;
  D.tor_nb = e->tor_nb; D.tor_nb_cnt = e->tor_nb_cnt;
  D.stats = e->stats_dev;
  const BatchGroups G = batch_groups(e, e->cfg.max_batch);   // pointers only: independent of B
  const GraphDyn& gd = e->gd;
  const FinGroup f_ll = fin_group(G.ll, gd.start_ll, gd.cnt_ll), f_lr = fin_group(G.lr, gd.start_lr, gd.cnt_lr);
  const FinGroup f_rl = fin_group(G.rl, gd.start_rl, gd.cnt_rl), f_rr = fin_group(G.rr, e->rr_start, e->rr_cnt);
  const FinGroup f_rr_shared = fin_group(G.rr_shared, e->rr_start, e->rr_cnt, e->gs.Nr);
  D.fin_lig.n_groups = 2; D.fin_lig.g[0] = f_ll; D.fin_lig.g[1] = f_lr;
  D.fin_rec.n_groups = 2; D.fin_rec.g[0] = f_rr; D.fin_rec.g[1] = f_rl;
  D.fin_rec_shared.n_groups = 2; D.fin_rec_shared.g[0] = f_rr_shared; D.fin_rec_shared.g[1] = f_rl;
#ifdef CBD_EXPERIMENTS
  {   // bf16 role split: + the second 0e slice (columns [0, NS) only, not counted in the degree) of every cross / receptor group
    auto second = [&](const FinGroup& f, int g) {
      FinGroup b = f;
      b.first_sum = f.first_sum + e->piece_b_off[g]; b.last_sum = f.last_sum + e->piece_b_off[g]; b.run_acc = f.run_acc + e->piece_b_off[g];
      b.deg_weight = 0; b.col_hi = NS;
      return b;
    };
    D.fin_lig_r.n_groups = 3; D.fin_lig_r.g[0] = f_ll; D.fin_lig_r.g[1] = f_lr; D.fin_lig_r.g[2] = second(f_lr, 1);
    D.fin_rec_r.n_groups = 4; D.fin_rec_r.g[0] = f_rr; D.fin_rec_r.g[1] = f_rl; D.fin_rec_r.g[2] = second(f_rr, 2); D.fin_rec_r.g[3] = second(f_rl, 3);
    D.fin_rec_shared_r.n_groups = 3; D.fin_rec_shared_r.g[0] = f_rr_shared; D.fin_rec_shared_r.g[1] = f_rl; D.fin_rec_shared_r.g[2] = second(f_rl, 3);
  }
#endif
  ConvGroupH sl[EMB_SLICES];
  emb_slices(e, G.ll, 0, sl);   // the piece buffers of the slices do not depend on the layer
  D.fin_emb.n_groups = EMB_SLICES;
  for (int k = 0; k < EMB_SLICES; ++k) {
    D.fin_emb.g[k] = fin_group(sl[k], gd.start_ll, gd.cnt_ll);
    D.fin_emb.g[k].deg_weight = k == 0 ? 1 : 0;
    D.fin_emb.g[k].col_hi = sl[k].vec_on ? 0 : NS;      // a 0e-only slice holds (the bf16 kernel: writes) the scalar columns only
  }
}

// Per-call part of the descriptor -> device (stream ordered).
static int push_desc(cbd_engine* e, int B, float* pos, float* tr, float* rot, float* tor, const float* ztr, const float* zrot,
                     const float* ztor, hipStream_t s) {
  PoseBatch& D = e->desc_h;
  D.B = B;
  D.cap_ll = B * e->cap_ll_per_sample;
  D.cap_x = B * e->gs.Nl * e->gs.Nr;
  D.gd.pos = pos;
  D.tr_out = tr; D.rot_out = rot; D.tor_out = tor;
  D.z_tr = ztr; D.z_rot = zrot; D.z_tor = ztor;
  e->last_B = B;
  HIPCHK(launch_set_desc(D, e->desc_dev, s));
  return 0;
}

// One score-model forward for the pose batches of n engines (one complex each, same weights) in shared launches.  The scores
// land in every batch's tr_out / rot_out / tor_out (descriptor).  sigma_emb_dev: device pointer to this step's 32-float embedding.
static int forward_multi(cbd_engine* const* E, int n, const cbd_step& st, const float* sigma_emb_dev, hipStream_t s) {
  cbd_engine* e0 = E[0];
  const PoseBatch* descs[MAX_COSCHED];
  BatchGroups G[MAX_COSCHED];
  int Bk[MAX_COSCHED];
  for (int k = 0; k < n; ++k) {
    descs[k] = E[k]->desc_dev;
    Bk[k] = E[k]->desc_h.B;
    G[k] = batch_groups(E[k], Bk[k]);
    E[k]->dbg_snap.clear();
  }
  auto nl = [&](int k) { return Bk[k] * E[k]->gs.Nl; };
  auto nr = [&](int k) { return Bk[k] * E[k]->gs.Nr; };
  const bool dbg = n == 1 && e0->keep_debug;
  const StepVectors& sv = e0->sv;   // functions of the diffusion time and the (shared) weights only: one set for all batches
  HIPCHK(launch_step_prep(e0->sw, sv, sigma_emb_dev, s));
  // ---- fork: everything that depends only on the diffusion time runs on the side stream, concurrently with the
  //      pose-dependent graph construction and ligand embedding: receptor rows (static embedding + sigma embedding,
  //      score_model.py:323-326) and the receptor->receptor messages of interaction layer 0.  Those messages read only
  //      receptor features and shared edge attributes, so they are IDENTICAL for the B samples of a complex: they are
  //      computed once (Err edges instead of B*Err) and added to every sample's sum by the finalize kernel (node_mod = Nr).
  HIPCHK(hipEventRecord(e0->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(e0->side, e0->ev_fork, 0));
  {
    static const bool no_side = getenv("CBD_NO_SIDE") != nullptr;   // diagnostic: keep the time-only work on the main stream
    hipStream_t ss = no_side ? s : e0->side;
    const Multi mt = make_multi(n, descs, [&](int k) { return (nr(k) * NODE_STRIDE + 255) / 256 + (E[k]->gs.Err * 32 + 255) / 256; });
    HIPCHK(launch_rec_time_init(mt, sv.rec_sigma_emb, 1, ss));
    ConvJob jobs[MAX_COSCHED];
    for (int k = 0; k < n; ++k) {
      ConvJob& J = jobs[k];
      J.e = E[k]; J.n_groups = 1; J.g[0] = G[k].rr_shared; J.caps[0] = E[k]->gs.Err; J.widx[0] = 2; J.node_in = E[k]->X1;
    }
    CHK(run_conv(e0->conv[0], jobs, n, ss, true));
    HIPCHK(hipEventRecord(e0->ev_join, ss));
  }
  const float lig_r = e0->cfg.lig_max_radius;
  const int lig_cap = e0->cfg.lig_radius_cap;
  HIPCHK(launch_graph_count(make_multi(n, descs, [&](int k) { return nl(k) + nr(k); }), lig_r, lig_cap, st.cross_cutoff, s));
  HIPCHK(launch_graph_scan(make_multi(n, descs, [&](int) { return 3; }), s));
  HIPCHK(launch_graph_fill(make_multi(n, descs, nl), make_multi(n, descs, nr), lig_r, lig_cap, st.cross_cutoff, s));
  {
    EdgeMlpArgs ma{};
    const EdgeMlp mll = make_mlp(e0->m_lig_edge, sv.ll_part), mlr = make_mlp(e0->m_cross, sv.lr_part);
    for (int k = 0; k < n; ++k) {
      const GraphDyn& gd = E[k]->gd;
      ma.seg[ma.n++] = EdgeSeg{mll, gd.ll_dist, gd.ll_bond4, gd.counts + 0, E[k]->desc_h.cap_ll, E[k]->ll_attr};
      ma.seg[ma.n++] = EdgeSeg{mlr, gd.lr_dist, nullptr, gd.counts + 1, E[k]->desc_h.cap_x, E[k]->lr_attr};
    }
    HIPCHK(launch_edge_mlp(ma, s));
  }
  HIPCHK(launch_lig_node_init(make_multi(n, descs, [&](int k) { return (nl(k) * NODE_STRIDE + 255) / 256; }), sv.lig_node_c, 0, s));
  if (dbg) snap(e0, "lig_node_emb0", e0->X0, (size_t)nl(0) * NODE_STRIDE, s);

  const Multi m_lig_nodes = make_multi(n, descs, [&](int k) { return (nl(k) * NODE_STRIDE + 255) / 256; });
  const Multi m_all_nodes = make_multi(n, descs, [&](int k) { return ((nl(k) + nr(k)) * NODE_STRIDE + 255) / 256; });
  int xi = 0;   // ligand embedding ping-pong: X0 -> X1 -> X0 -> X1 ; the interaction layers read X1 first
  static const char* emb_names[3] = {"lig_emb_0", "lig_emb_1", "lig_emb_2"};
  ConvJob jobs[MAX_COSCHED];
  for (int l = 0; l < 3; ++l) {   // ligand embedding layers on the ligand graph only (score_model.py:289-293)
    const ConvLayerDev& L = e0->lig_emb[l];
    for (int k = 0; k < n; ++k) {
      ConvJob& J = jobs[k];
      J = ConvJob{};
      J.e = E[k]; J.n_groups = EMB_SLICES; J.node_in = E[k]->desc_h.X[xi];
      ConvGroupH sl[EMB_SLICES];
      emb_slices(E[k], G[k].ll, l, sl);
      for (int q = 0; q < EMB_SLICES; ++q) { J.g[q] = sl[q]; J.caps[q] = E[k]->desc_h.cap_ll; J.widx[q] = 0; }
    }
    CHK(run_conv(L, jobs, n, s, false));
    HIPCHK(launch_conv_finalize_multi(m_lig_nodes, FIN_EMB, xi, xi ^ 1, L.bn_scale, L.bn_mean, L.bn_bias, in_level_dim(L.in_level),
                                      out_level_dim(L.out_level), s));
    xi ^= 1;
    if (dbg) snap(e0, emb_names[l], e0->desc_h.X[xi], (size_t)nl(0) * NODE_STRIDE, s);
  }
  // ---- join: X[xi] (== X1) now holds the embedded ligand rows (main stream) and the receptor rows (side stream)
  HIPCHK(hipStreamWaitEvent(s, e0->ev_join, 0));
  static const char* conv_names[5] = {"conv_0", "conv_1", "conv_2", "conv_3", "conv_4"};
  const bool roles = e0->use_bf16 == 1 && e0->bf16_roles != 0;
  for (int l = 0; l < 5; ++l) {   // interaction layers on the joint graph (score_model.py:365-374)
    const ConvLayerDev& L = e0->conv[l];
    for (int k = 0; k < n; ++k) {
      ConvJob& J = jobs[k];
      J = ConvJob{};
      J.e = E[k]; J.node_in = E[k]->desc_h.X[xi];
      const int cap_ll = E[k]->desc_h.cap_ll, cap_x = E[k]->desc_h.cap_x, cap_rr = Bk[k] * E[k]->gs.Err;
#ifdef CBD_EXPERIMENTS
      if (roles) {
        // three virtual slices per cross / receptor group: 0e tiles [0, h0), [h0, t0e) and the vector blocks; what each costs a CU is
        // a third of the weight stream, and the persistent kernel keeps a 0e slice's tiles in LDS (tp_conv_bf16p.hip)
        const ConvShape S3 = conv_shape(3, 3);
        const int h0 = (S3.t0e + 1) / 2;
        J.n_groups = 0;
        auto add = [&](const ConvGroupH& g, int cap, int w) { J.g[J.n_groups] = g; J.caps[J.n_groups] = cap; J.widx[J.n_groups] = w; ++J.n_groups; };
        auto add3 = [&](const ConvGroupH& g, int gi, int cap, int w) {
          ConvGroupH a = g, b = g, c = g;
          a.i0e_lo = 0; a.i0e_hi = h0; a.vec_on = 0;
          b.i0e_lo = h0; b.i0e_hi = S3.t0e; b.vec_on = 0;
          b.first_sum += E[k]->piece_b_off[gi]; b.last_sum += E[k]->piece_b_off[gi]; b.run_acc += E[k]->piece_b_off[gi];
          c.i0e_lo = S3.t0e; c.i0e_hi = S3.t0e; c.vec_on = 1;
          add(a, cap, w); add(b, cap, w); add(c, cap, w);
        };
        add(G[k].ll, cap_ll, 0);
        add3(G[k].lr, 1, cap_x, 1);
        if (l >= 1 && l < 4) add3(G[k].rr, 2, cap_rr, 2);
        if (l < 4) add3(G[k].rl, 3, cap_x, 3);
      } else
#endif
      if (l == 0) {          // the receptor->receptor group of layer 0 is the shared one computed on the side stream
        J.n_groups = 3;
        J.g[0] = G[k].ll; J.g[1] = G[k].lr; J.g[2] = G[k].rl;
        J.caps[0] = cap_ll; J.caps[1] = cap_x; J.caps[2] = cap_x;
        J.widx[0] = 0; J.widx[1] = 1; J.widx[2] = 3;
      } else if (l < 4) {
        J.n_groups = 4;
        J.g[0] = G[k].ll; J.g[1] = G[k].lr; J.g[2] = G[k].rr; J.g[3] = G[k].rl;
        J.caps[0] = cap_ll; J.caps[1] = cap_x; J.caps[2] = cap_rr; J.caps[3] = cap_x;
      } else {
        J.n_groups = 2;
        J.g[0] = G[k].ll; J.g[1] = G[k].lr;
        J.caps[0] = cap_ll; J.caps[1] = cap_x;
      }
    }
    CHK(run_conv(L, jobs, n, s, false));
    const int kind = (l == 0 ? FIN_FIRST : l < 4 ? FIN_MID : FIN_LAST) + (roles ? FIN_FIRST_R - FIN_FIRST : 0);   // receptor rows of the last layer are never read again (quirk 3)
    HIPCHK(launch_conv_finalize_multi(l < 4 ? m_all_nodes : m_lig_nodes, kind, xi, xi ^ 1, L.bn_scale, L.bn_mean, L.bn_bias,
                                      in_level_dim(L.in_level), out_level_dim(L.out_level), s));
    xi ^= 1;
    if (dbg) {
      snap(e0, conv_names[l], e0->desc_h.X[xi], (size_t)nl(0) * NODE_STRIDE, s);
      if (l < 4) snap(e0, (std::string(conv_names[l]) + "_rec").c_str(), e0->desc_h.X[xi] + (size_t)e0->gs.rec_off * NODE_STRIDE, (size_t)nr(0) * NODE_STRIDE, s);
    }
  }
  HIPCHK(launch_center_head(e0->ch, sv, make_multi(n, descs, nl), make_multi(n, descs, [&](int k) { return Bk[k]; }), xi, st.tr_sigma,
                            st.rot_score_norm, s));
  if (!e0->cfg.no_torsion) {
    const Multi mb = make_multi(n, descs, [&](int k) { return Bk[k] * E[k]->gs.R; });
    HIPCHK(launch_bond_nb(mb, lig_r, 32, s));
    HIPCHK(launch_bond_conv(e0->bh, mb, xi, e0->bond_stream, st.tor_score_norm_sqrt, s));
  }
  for (int k = 0; k < n; ++k) {
    cbd_engine* e = E[k];
    e->dbg.clear();
    e->dbg["center_mean"] = {e->dbg_global, (size_t)Bk[k] * 12};
    e->dbg["tor_feat"] = {e->dbg_torfeat, (size_t)Bk[k] * e->gs.R * 64};
    e->dbg["rec_node_static"] = {e->rec_static, (size_t)e->gs.Nr * NODE_STRIDE};
    e->dbg["rec_sigma_emb"] = {sv.rec_sigma_emb, 32};
    e->dbg["lig_node_final"] = {e->desc_h.X[xi], (size_t)nl(k) * NODE_STRIDE};
    e->dbg["ll_attr"] = {e->ll_attr, 0};   // count filled at fetch time
    e->dbg["lr_attr"] = {e->lr_attr, 0};
  }
  return 0;
}

static int check_batch(cbd_engine* e, int B) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  if (!e->weights_ready || !e->complex_ready) return fail(CBD_ERR_STATE, "weights and complex must be set first");
  if (B <= 0) return fail(CBD_ERR_ARG, "batch must be positive");
  if (B > e->cfg.max_batch) return fail(CBD_ERR_CAPACITY, "batch %d exceeds max_batch %d", B, e->cfg.max_batch);
  return 0;
}

static void collect_range(cbd_engine* e, const cbd_engine::EvPool& pool, size_t lo, size_t hi) {
  for (size_t i = lo; i < hi; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(pool[i].second) == hipSuccess && hipEventElapsedTime(&ms, pool[i].first, pool[i].second) == hipSuccess) {
      e->t_total_ms += ms;
      e->t_n += 1;
    }
  }
}

// Eager launches: every launch since the last collection has its own event pair.  Captured graphs: the pairs are nodes of the graph
// and hold the durations of its most recent replay; a graph is read before it is replayed again (sample_impl) and here.
static void collect_timing(cbd_engine* e) {
  collect_range(e, e->ev_pool, 0, e->ev_used);
  e->ev_used = 0;
  for (auto& g : e->graphs)
    if (g.replayed) { collect_range(e, e->gev_pool, g.ev_lo, g.ev_hi); g.replayed = false; }
}

int cbd_score(cbd_engine* e, int32_t B, const float* pos_dev, const cbd_step* step, float* tr_dev, float* rot_dev, float* tor_dev,
              void* stream) {
  CHK(check_batch(e, B));
  if (!pos_dev || !step || !tr_dev || !rot_dev) return fail(CBD_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  e->sync_all = true;
  HIPCHK(hipMemcpyAsync(e->sigma_emb_dev, step->sigma_emb, 64 * sizeof(float), hipMemcpyHostToDevice, s));   // sigma_emb | sigma_emb_t (adjacent)
  CHK(push_desc(e, B, const_cast<float*>(pos_dev), tr_dev, rot_dev, tor_dev ? tor_dev : e->tor_out, nullptr, nullptr, nullptr, s));
  cbd_engine* E[1] = {e};
  CHK(forward_multi(E, 1, *step, e->sigma_emb_dev, s));
  return 0;
}

int cbd_modify_conformer(cbd_engine* e, int32_t B, float* pos_dev, const float* tr_dev, const float* rot_dev, const float* tor_dev,
                         void* stream) {
  if (!e || !e->complex_ready) return fail(CBD_ERR_STATE, "complex must be set first");
  if (B <= 0 || B > e->cfg.max_batch || !pos_dev || !tr_dev || !rot_dev) return fail(CBD_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  e->sync_all = true;
  // the given updates take the place of the scores (use_coefs = 0): descriptor outputs point at them
  CHK(push_desc(e, B, pos_dev, const_cast<float*>(tr_dev), const_cast<float*>(rot_dev), const_cast<float*>(tor_dev), nullptr, nullptr,
                nullptr, s));
  const PoseBatch* d[1] = {e->desc_dev};
  HIPCHK(launch_pose_update(make_multi(1, d, [&](int) { return B; }), 0, SdeCoefs{}, 0, tor_dev != nullptr && e->gs.R > 0, e->gs.Nl, s));
  return 0;
}

// The step loop of n co-scheduled batches (n = 1: cbd_sample).
static int sample_impl(int n, cbd_engine* const* E, const int32_t* B, int32_t S, const cbd_step* steps, float* const* pos_dev,
                       const float* const* noise_tr, const float* const* noise_rot, const float* const* noise_tor, float* scores_out,
                       hipStream_t s) {
  cbd_engine* e0 = E[0];
  auto nz = [](const float* const* a, int k) { return a ? a[k] : nullptr; };
  if (S > e0->sigma_cap) {
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(e0->bpool.alloc(&e0->sigma_emb_dev, (size_t)S * 64));
    e0->sigma_cap = S;
  }
  std::vector<float> se((size_t)S * 64);
  for (int i = 0; i < S; ++i) std::memcpy(se.data() + (size_t)i * 64, steps[i].sigma_emb, 64 * sizeof(float));
  HIPCHK(hipMemcpyAsync(e0->sigma_emb_dev, se.data(), se.size() * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));   // `se` goes out of scope; also orders the upload before the loop
  int max_nl = 0;
  bool any_tors = false;
  for (int k = 0; k < n; ++k) {
    max_nl = std::max(max_nl, E[k]->gs.Nl);
    any_tors = any_tors || (!E[k]->cfg.no_torsion && E[k]->gs.R > 0);
  }
  const PoseBatch* descs[MAX_COSCHED];
  for (int k = 0; k < n; ++k) descs[k] = E[k]->desc_dev;
  const Multi m_samples = make_multi(n, descs, [&](int k) { return (int)B[k]; });
  auto run_steps = [&](float* scores) -> int {
    for (int i = 0; i < S; ++i) {
      const cbd_step& st = steps[i];
      CHK(forward_multi(E, n, st, e0->sigma_emb_dev + (size_t)i * 64, s));
      if (scores) {   // n == 1 only (tests)
        const int R = e0->gs.R, B0 = B[0];
        float* o = scores + (size_t)i * B0 * (6 + R);
        HIPCHK(hipMemcpyAsync(o, e0->tr_out, (size_t)B0 * 3 * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(o + B0 * 3, e0->rot_out, (size_t)B0 * 3 * 4, hipMemcpyDeviceToDevice, s));
        if (any_tors) HIPCHK(hipMemcpyAsync(o + B0 * 6, e0->tor_out, (size_t)B0 * R * 4, hipMemcpyDeviceToDevice, s));
      }
      const SdeCoefs cf{st.tr_score_coef, st.tr_noise_coef, st.rot_score_coef, st.rot_noise_coef, st.tor_score_coef, st.tor_noise_coef};
      HIPCHK(launch_pose_update(m_samples, i, cf, 1, any_tors ? 1 : 0, max_nl, s));
    }
    return 0;
  };
  // Kernel timing under a graph: the event pairs are explicit event-record nodes of the graph (record_event).
  bool graph_ok = e0->use_graph && !scores_out;
  for (int k = 0; k < n; ++k) graph_ok = graph_ok && !E[k]->keep_debug;
  if (!graph_ok) {
    for (int k = 0; k < n; ++k)
      CHK(push_desc(E[k], B[k], pos_dev[k], E[k]->tr_out, E[k]->rot_out, E[k]->tor_out, nz(noise_tr, k), nz(noise_rot, k), nz(noise_tor, k), s));
    CHK(run_steps(scores_out));
    for (int k = 0; k < n; ++k) { HIPCHK(hipEventRecord(E[k]->ev_last, s)); E[k]->last_used = true; }
    return 0;   // asynchronous: kernel-timing events are collected when cbd_kernel_timing is queried
  }
  // ---- the whole S-step loop of all batches as ONE hipGraph launch (static capacities + device-side edge counts make every launch
  //      shape independent of the data).  Inputs are staged into engine-owned buffers so that the instantiated graph can be replayed
  //      for every group of batches with the same engines, complexes, batch sizes and schedule.
  hipStream_t user = s;
  if (!user) {   // the legacy default stream cannot be captured: run on the engine's own stream, ordered after/before it
    HIPCHK(hipEventRecord(e0->ev_a, user));
    HIPCHK(hipStreamWaitEvent(e0->own, e0->ev_a, 0));
    s = e0->own;
  }
  std::string key(reinterpret_cast<const char*>(steps), sizeof(cbd_step) * (size_t)S);
  for (int k = 0; k < n; ++k) {
    cbd_engine* e = E[k];
    const int Bm = e->cfg.max_batch, Nl = e->gs.Nl, R = e->gs.R;
    if (!e->g_pos || e->g_S_cap < S) {
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(e->bpool.alloc(&e->g_pos, (size_t)Bm * Nl * 3));
      HIPCHK(e->bpool.alloc(&e->g_ztr, (size_t)S * Bm * 3)); HIPCHK(e->bpool.alloc(&e->g_zrot, (size_t)S * Bm * 3));
      HIPCHK(e->bpool.alloc(&e->g_ztor, (size_t)S * Bm * std::max(R, 1)));
      e->g_S_cap = S;
      e->complex_gen = cbd_engine::next_gen();
    }
    char buf[96];
    snprintf(buf, sizeof buf, "|%p:%llu:%d:%d:%d:%d", (void*)e, (unsigned long long)e->complex_gen, (int)B[k], e->use_bf16 + 8 * e->bf16_roles + 32 * e->bf16_stat,
             (nz(noise_tr, k) != nullptr) + 2 * (nz(noise_rot, k) != nullptr) + 4 * (nz(noise_tor, k) != nullptr), (int)e->timing);
    key += buf;
  }
  // descriptors point at the staging buffers; refreshed before every launch (an eager call in between may have re-pointed them)
  for (int k = 0; k < n; ++k) {
    cbd_engine* e = E[k];
    CHK(push_desc(e, B[k], e->g_pos, e->tr_out, e->rot_out, e->tor_out, nz(noise_tr, k) ? e->g_ztr : nullptr,
                  nz(noise_rot, k) ? e->g_zrot : nullptr, nz(noise_tor, k) ? e->g_ztor : nullptr, s));
  }
  hipGraphExec_t exec = nullptr;
  for (auto& g : e0->graphs)
    if (g.key == key) {
      if (g.replayed && g.ev_hi > g.ev_lo) { collect_range(e0, e0->gev_pool, g.ev_lo, g.ev_hi); }   // waits for the previous replay
      exec = g.exec; g.replayed = true;
    }
  if (!exec) {
    hipGraph_t graph = nullptr;
    const size_t ev_lo = e0->gev_used;
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = run_steps(nullptr);
    const hipError_t ce = hipStreamEndCapture(s, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    HIPCHK(ce);
    HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(graph));
    if (e0->graphs.size() >= 8) {   // small cache: the oldest entry goes
      HIPCHK(hipStreamSynchronize(s));
      (void)hipGraphExecDestroy(e0->graphs.front().exec);
      e0->graphs.erase(e0->graphs.begin());
    }
    e0->graphs.push_back({key, exec, ev_lo, e0->gev_used, true});
  }
  for (int k = 0; k < n; ++k) {
    cbd_engine* e = E[k];
    const int Nl = e->gs.Nl, R = e->gs.R, Bk = B[k];
    HIPCHK(hipMemcpyAsync(e->g_pos, pos_dev[k], (size_t)Bk * Nl * 3 * 4, hipMemcpyDeviceToDevice, s));
    if (nz(noise_tr, k)) HIPCHK(hipMemcpyAsync(e->g_ztr, noise_tr[k], (size_t)S * Bk * 3 * 4, hipMemcpyDeviceToDevice, s));
    if (nz(noise_rot, k)) HIPCHK(hipMemcpyAsync(e->g_zrot, noise_rot[k], (size_t)S * Bk * 3 * 4, hipMemcpyDeviceToDevice, s));
    if (nz(noise_tor, k) && R > 0) HIPCHK(hipMemcpyAsync(e->g_ztor, noise_tor[k], (size_t)S * Bk * R * 4, hipMemcpyDeviceToDevice, s));
  }
  HIPCHK(hipGraphLaunch(exec, s));
  for (int k = 0; k < n; ++k)
    HIPCHK(hipMemcpyAsync(pos_dev[k], E[k]->g_pos, (size_t)B[k] * E[k]->gs.Nl * 3 * 4, hipMemcpyDeviceToDevice, s));
  for (int k = 0; k < n; ++k) { HIPCHK(hipEventRecord(E[k]->ev_last, s)); E[k]->last_used = true; }      // what an asynchronous cbd_set_complex of these engines waits for
  if (!user) {
    HIPCHK(hipEventRecord(e0->ev_b, s));
    HIPCHK(hipStreamWaitEvent(user, e0->ev_b, 0));
  }
  return 0;
}

int cbd_sample(cbd_engine* e, int32_t B, int32_t S, const cbd_step* steps, float* pos_dev, const float* noise_tr,
               const float* noise_rot, const float* noise_tor, float* scores_out, void* stream) {
  CHK(check_batch(e, B));
  if (S <= 0 || !steps || !pos_dev) return fail(CBD_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  cbd_engine* E[1] = {e};
  const int32_t Bs[1] = {B};
  float* ps[1] = {pos_dev};
  const float* tr[1] = {noise_tr};
  const float* rot[1] = {noise_rot};
  const float* tor[1] = {noise_tor};
  return sample_impl(1, E, Bs, S, steps, ps, tr, rot, tor, scores_out, reinterpret_cast<hipStream_t>(stream));
}

int cbd_sample_multi(int32_t n, cbd_engine* const* engines, const int32_t* B, int32_t S, const cbd_step* steps, float* const* pos_dev,
                     const float* const* noise_tr, const float* const* noise_rot, const float* const* noise_tor, void* stream) {
  if (n < 1 || n > MAX_COSCHED || !engines || !B || !pos_dev) return fail(CBD_ERR_ARG, "1..%d engines are required", MAX_COSCHED);
  if (S <= 0 || !steps) return fail(CBD_ERR_ARG, "bad argument");
  cbd_engine* e0 = engines[0];
  for (int k = 0; k < n; ++k) {
    cbd_engine* e = engines[k];
    if (!e) return fail(CBD_ERR_ARG, "null engine");
    for (int q = 0; q < k; ++q)
      if (engines[q] == e) return fail(CBD_ERR_ARG, "distinct engines are required");
    if (!pos_dev[k]) return fail(CBD_ERR_ARG, "null pose buffer");
    if (e->cfg.device != e0->cfg.device) return fail(CBD_ERR_ARG, "co-scheduled engines must live on the same device");
    if (e->use_bf16 != e0->use_bf16 || int(e->bf16_roles) - int(e0->bf16_roles) != 0 || e->bf16_stat != e0->bf16_stat) return fail(CBD_ERR_ARG, "co-scheduled engines must use the same operand precision");
    if (e->cfg.no_torsion != e0->cfg.no_torsion || e->cfg.lig_max_radius != e0->cfg.lig_max_radius ||
        e->cfg.lig_radius_cap != e0->cfg.lig_radius_cap)
      return fail(CBD_ERR_ARG, "co-scheduled engines must share one model configuration");
    if (!e->weights_ready || !e0->weights_ready || e->conv[0].wstream[0] != e0->conv[0].wstream[0])
      return fail(CBD_ERR_ARG, "co-scheduled engines must share one set of weights (cbd_share_weights)");
    CHK(check_batch(e, B[k]));
  }
  HIPCHK(hipSetDevice(e0->cfg.device));
  return sample_impl(n, engines, B, S, steps, pos_dev, noise_tr, noise_rot, noise_tor, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int cbd_sample_pair(cbd_engine* e0, cbd_engine* e1, int32_t B0, int32_t B1, int32_t S, const cbd_step* steps, float* pos0_dev,
                    const float* noise_tr0, const float* noise_rot0, const float* noise_tor0, float* pos1_dev,
                    const float* noise_tr1, const float* noise_rot1, const float* noise_tor1, void* stream) {
  cbd_engine* es[2] = {e0, e1};
  const int32_t Bs[2] = {B0, B1};
  float* ps[2] = {pos0_dev, pos1_dev};
  const float* tr[2] = {noise_tr0, noise_tr1};
  const float* rot[2] = {noise_rot0, noise_rot1};
  const float* tor[2] = {noise_tor0, noise_tor1};
  if (!e0 || !e1 || e0 == e1) return fail(CBD_ERR_ARG, "two distinct engines are required");
  return cbd_sample_multi(2, es, Bs, S, steps, ps, tr, rot, tor, stream);
}

int cbd_set_option(cbd_engine* e, const char* name, int64_t value) {
  if (!e || !name) return fail(CBD_ERR_ARG, "null argument");
  const std::string k(name);
  if (k == "graph") {
    e->use_graph = value != 0;
    if (!e->use_graph) drop_graphs(e);
    return 0;
  }
  if (k == "bf16") {   // captured graphs bake the kernel choice in: drop them
    if (value != 0) e->use_bf16 = 1; else if (e->use_bf16 == 1) e->use_bf16 = 0;
    drop_graphs(e);
    return 0;
  }
  if (k == "bf16_roles") {   // bf16 only: cross / receptor groups as three tile slices per layer (captured graphs bake it in)
#ifdef CBD_EXPERIMENTS
    e->bf16_roles = (int)std::max<long long>(0, std::min<long long>(2, value));
    drop_graphs(e);
    return 0;
#else
    if (value == 0) return 0;
    return fail(CBD_ERR_ARG, "bf16_roles is an experiment of the diagnostic library (tools/diag_lib.py), not part of libcbdock.so");
#endif
  }
  if (k == "async_setup") {   // cbd_set_complex on a stream of its own, waiting only for this engine's last cbd_sample* launches (see cbd_engine)
    e->async_setup = value != 0;
    return 0;
  }
  if (k == "bf16_stationary") {   // bf16 only: register-stationary kernel for the 74 -> 74 layers (captured graphs bake it in)
    e->bf16_stat = value != 0;
    drop_graphs(e);
    return 0;
  }
  if (k == "f32_split") {   // fp32 operands as three bf16 planes on the bf16 matrix cores (OpsBf16x3)
    if (value != 0) e->use_bf16 = 2; else if (e->use_bf16 == 2) e->use_bf16 = 0;
    drop_graphs(e);
    return 0;
  }
  return fail(CBD_ERR_ARG, "unknown option '%s'", name);
}

int cbd_share_weights(cbd_engine* dst, cbd_engine* src) {
  if (!dst || !src || dst == src) return fail(CBD_ERR_ARG, "bad argument");
  if (!src->weights_ready) return fail(CBD_ERR_STATE, "source engine has no finalized weights");
  if (dst->cfg.device != src->cfg.device) return fail(CBD_ERR_ARG, "engines live on different devices");
  HIPCHK(hipSetDevice(dst->cfg.device));
  dst->wpool.release();          // dst becomes a non-owning view of src's device weights (src must outlive dst)
  for (int l = 0; l < 3; ++l) { dst->rec_emb[l] = src->rec_emb[l]; dst->lig_emb[l] = src->lig_emb[l]; }
  for (int l = 0; l < 5; ++l) dst->conv[l] = src->conv[l];
  dst->m_lig_edge = src->m_lig_edge; dst->m_cross = src->m_cross; dst->m_rec_edge = src->m_rec_edge;
  dst->m_final_edge = src->m_final_edge; dst->m_center = src->m_center;
  dst->sw = src->sw; dst->ch = src->ch; dst->bh = src->bh; dst->bond_stream = src->bond_stream;
  dst->rec_emb_table = src->rec_emb_table; dst->rec_node_w = src->rec_node_w; dst->rec_node_b = src->rec_node_b;
  dst->lig_node_w_host = src->lig_node_w_host; dst->lig_emb_tables = src->lig_emb_tables;
  dst->weights_ready = true;
  dst->complex_ready = false;
  return 0;
}

int cbd_recompute_receptor(cbd_engine* e, void* stream) {
  if (!e || !e->complex_ready) return fail(CBD_ERR_STATE, "complex must be set first");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->sync_all = true;
  CHK(embed_receptor(e, reinterpret_cast<hipStream_t>(stream)));
  return 0;
}

int cbd_stats(cbd_engine* e, int32_t reset, uint64_t out[4]) {
  if (!e || !e->complex_ready) return fail(CBD_ERR_STATE, "complex must be set first");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  if (out) HIPCHK(hipMemcpy(out, e->stats_dev, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(e->stats_dev, 0, 4 * sizeof(unsigned long long)));
  return 0;
}

int64_t cbd_debug_fetch(cbd_engine* e, const char* name, float* out, int64_t capacity) {
  if (!e || !name) return fail(CBD_ERR_ARG, "null argument");
  const std::string k(name);
  if (k == "enable") { e->keep_debug = true; return 0; }
  if (k == "disable") { e->keep_debug = false; e->dbg_snap.clear(); return 0; }
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& p : e->dbg_snap)
    if (p.first == k) {
      if ((int64_t)p.second.size() > capacity) return fail(CBD_ERR_ARG, "capacity too small for '%s' (%zu)", name, p.second.size());
      std::memcpy(out, p.second.data(), p.second.size() * 4);
      return (int64_t)p.second.size();
    }
  if (k == "conv_clock_ghz") {   // median in-kernel shader clock of the last tp_conv<3,3> launch (diagnostic build)
    if (!e->stamps_dev || capacity < 3) return fail(CBD_ERR_ARG, "stamps not enabled: phase stamps exist in the diagnostic library only (tools/diag_lib.py)");
    std::vector<unsigned long long> h(8192 * 8);
    if (hipMemcpy(h.data(), e->stamps_dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(CBD_ERR_HIP, "memcpy failed");
    std::vector<double> ghz, dur, pro, g1, tiles, fin, g1a;
    for (int i = 0; i < 8192; ++i) {
      const unsigned long long* q = h.data() + 8 * i;
      const double dt = (double)(q[2] - q[0]), dr = (double)(q[3] - q[1]);
      if (q[3] && dr > 0) {
        ghz.push_back(dt / dr * 0.1); dur.push_back(dr * 10.0);   // memrealtime ticks at 100 MHz
        pro.push_back((double)(q[4] - q[0])); g1.push_back((double)(q[5] - q[4])); tiles.push_back((double)(q[6] - q[5]));
        fin.push_back((double)(q[2] - q[6]));
        g1a.push_back((double)(q[7] - q[4]));
      }
    }
    if (ghz.empty()) return fail(CBD_ERR_STATE, "no stamps recorded");
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    if (capacity >= 7) { out[3] = (float)med(pro); out[4] = (float)med(g1); out[5] = (float)med(tiles); out[6] = (float)med(fin); }
    if (capacity >= 8) out[7] = (float)med(g1a);
    std::sort(ghz.begin(), ghz.end()); std::sort(dur.begin(), dur.end());
    out[0] = (float)ghz[ghz.size() / 2]; out[1] = (float)dur[dur.size() / 2]; out[2] = (float)ghz.size();
    return capacity >= 8 ? 8 : capacity >= 7 ? 7 : 3;
  }
#ifdef CBD_DIAG
  if (k == "conv_span_wg") {   // per workgroup of the last tp_conv64s launch (CBD_BF16_DIAG=5): start offset ns, lifetime ns, units, last role
    if (!e->stamps_dev || capacity < 4) return fail(CBD_ERR_ARG, "stamps not enabled (diagnostic library only, tools/diag_lib.py)");
    std::vector<unsigned long long> h(8192 * 8);
    if (hipMemcpy(h.data(), e->stamps_dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(CBD_ERR_HIP, "memcpy failed");
    unsigned long long t0 = ~0ull;
    for (int rec = 0; rec < 4096; rec += 4) {
      const unsigned long long* q = h.data() + 16 * (size_t)rec;
      if (q[3] && q[3] > q[1] && q[10]) t0 = std::min(t0, q[1]);
    }
    int64_t n = 0;
    for (int rec = 0; rec < 4096 && 4 * (n + 1) <= capacity; rec += 4) {
      const unsigned long long* q = h.data() + 16 * (size_t)rec;
      if (!q[3] || q[3] <= q[1] || q[10] == 0) continue;
      out[4 * n + 0] = (float)((double)(q[1] - t0) * 10.0);
      out[4 * n + 1] = (float)((double)(q[3] - q[1]) * 10.0);
      out[4 * n + 2] = (float)q[10];
      out[4 * n + 3] = (float)((double)q[11] + (double)(rec / 4) / 1024.0);      // role + workgroup index / 1024
      ++n;
    }
    return 4 * n;
  }
#endif
  if (k == "conv_clock_s") {   // per-wave phase clocks of the last tp_conv64s launch (CBD_BF16_DIAG=4): 4 waves x 10 floats (medians)
    if (!e->stamps_dev || capacity < 48) return fail(CBD_ERR_ARG, "stamps not enabled (diagnostic library only, tools/diag_lib.py) or capacity < 48");
    std::vector<unsigned long long> h(8192 * 8);
    if (hipMemcpy(h.data(), e->stamps_dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(CBD_ERR_HIP, "memcpy failed");
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    for (int w = 0; w < 4; ++w) {
      std::vector<double> col[10];
      for (int rec = w; rec < 4096; rec += 4) {
        const unsigned long long* q = h.data() + 16 * (size_t)rec;
        const double dt = (double)(q[2] - q[0]), dr = (double)(q[3] - q[1]);
        if (!q[3] || dr <= 0 || q[10] == 0) continue;
        col[0].push_back(dt); col[1].push_back(dt / dr * 0.1);
        for (int c = 0; c < 6; ++c) col[2 + c].push_back((double)q[4 + c]);
        col[8].push_back((double)q[10]); col[9].push_back(1.0);
      }
      for (int c = 0; c < 9; ++c) out[10 * w + c] = (float)med(col[c]);
      out[10 * w + 9] = (float)col[9].size();
    }
    {   // the launch as a whole (100 MHz real-time counter): span from the first start to the last end, lifetimes of the workgroups (wave 0)
      unsigned long long t0 = ~0ull, t1 = 0;
      std::vector<double> life, units;
      for (int rec = 0; rec < 4096; rec += 4) {
        const unsigned long long* q = h.data() + 16 * (size_t)rec;
        if (!q[3] || q[3] <= q[1] || q[10] == 0) continue;
        t0 = std::min(t0, q[1]); t1 = std::max(t1, q[3]);
        life.push_back((double)(q[3] - q[1]) * 10.0); units.push_back((double)q[10]);
      }
      std::sort(life.begin(), life.end());
      out[40] = life.empty() ? 0.f : (float)((double)(t1 - t0) * 10.0);
      out[41] = life.empty() ? 0.f : (float)life[life.size() / 2];
      out[42] = life.empty() ? 0.f : (float)life.back();
      out[43] = life.empty() ? 0.f : (float)life.front();
      out[44] = units.empty() ? 0.f : (float)*std::max_element(units.begin(), units.end());
      out[45] = units.empty() ? 0.f : (float)*std::min_element(units.begin(), units.end());
      out[46] = (float)life.size(); out[47] = 0.f;
      // per role (weight stream): median nanoseconds per unit over the workgroups whose LAST piece belongs to it and is a large one
      for (int r = 0; r < 4; ++r) {
        std::vector<double> per;
        for (int rec = 0; rec < 4096; rec += 4) {
          const unsigned long long* q = h.data() + 16 * (size_t)rec;
          if (!q[3] || q[3] <= q[1] || q[10] < 32 || (int)q[11] != r) continue;
          per.push_back((double)(q[3] - q[1]) * 10.0 / (double)q[10]);
        }
        if (capacity >= 56) { out[48 + 2 * r] = per.empty() ? 0.f : (float)med(per); out[49 + 2 * r] = (float)per.size(); }
      }
    }
    return capacity >= 56 ? 56 : 48;
  }
  auto it = e->dbg.find(k);
  if (it == e->dbg.end()) return fail(CBD_ERR_ARG, "unknown debug tensor '%s'", name);
  size_t n = it->second.second;
  if (k == "ll_attr" || k == "lr_attr") {
    int c[8];
    if (hipMemcpy(c, e->gd.counts, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return fail(CBD_ERR_HIP, "memcpy failed");
    n = (size_t)c[k == "ll_attr" ? 0 : 1] * 32;
  }
  if ((int64_t)n > capacity) return fail(CBD_ERR_ARG, "capacity too small for '%s' (%zu)", name, n);
  if (hipMemcpy(out, it->second.first, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(CBD_ERR_HIP, "memcpy failed");
  return (int64_t)n;
}

int cbd_last_edge_counts(cbd_engine* e, int64_t counts[5]) {
  if (!e || !e->complex_ready) return fail(CBD_ERR_STATE, "complex must be set first");
  (void)hipSetDevice(e->cfg.device);
  int c[8];
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(c, e->gd.counts, sizeof c, hipMemcpyDeviceToHost));
  for (int i = 0; i < 5; ++i) counts[i] = c[i];
  return 0;
}

int cbd_kernel_timing(cbd_engine* e, int32_t enable, int32_t reset, double* avg_ms, int64_t* n, double* total_ms) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  (void)hipSetDevice(e->cfg.device);
  if (e->ev_used || e->gev_used) { HIPCHK(hipDeviceSynchronize()); collect_timing(e); }
  if (reset) { e->t_total_ms = 0; e->t_n = 0; }
  e->timing = enable != 0;
  if (avg_ms) *avg_ms = e->t_n ? e->t_total_ms / (double)e->t_n : 0.0;
  if (n) *n = e->t_n;
  if (total_ms) *total_ms = e->t_total_ms;
  return 0;
}

// Host-only helper for the CPU tests: pack one FCBlock into the MFMA tile stream (no GPU needed).
// out must hold cbd_conv_stream_floats(in_level, out_level) floats.
int64_t cbd_conv_stream_floats(int32_t in_level, int32_t out_level) {
  return (int64_t)conv_stream_floats(conv_shape(in_level, out_level).ntiles);
}
int cbd_pack_conv_stream(int32_t in_level, int32_t out_level, const float* w1, const float* b1, const float* w2, const float* b2,
                         float* out) {
  if (in_level < 0 || in_level > 3 || out_level < 1 || out_level > 3) return fail(CBD_ERR_ARG, "bad level");
  std::vector<float> v = pack_conv_stream(in_level, out_level, w1, b1, w2, b2);
  std::memcpy(out, v.data(), v.size() * 4);
  return 0;
}

// the same for the layout the INFERENCE kernel reads (ConvShape::vmerged: one tile less where the vector blocks' tails fit one tile)
int64_t cbd_conv_stream_floats_infer(int32_t in_level, int32_t out_level) {
  return (int64_t)conv_stream_floats(conv_shape(in_level, out_level, true).ntiles);
}
int cbd_pack_conv_stream_infer(int32_t in_level, int32_t out_level, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* out) {
  if (in_level < 0 || in_level > 3 || out_level < 1 || out_level > 3) return fail(CBD_ERR_ARG, "bad level");
  std::vector<float> v = pack_conv_stream(in_level, out_level, w1, b1, w2, b2, true);
  std::memcpy(out, v.data(), v.size() * 4);
  return 0;
}

int cbd_symm_rmsd(int32_t B, int32_t N, int32_t K, const float* pos_dev, const float* ref_dev, const int32_t* idx_ref_dev,
                  const int32_t* idx_pos_dev, float* rmsd_out_dev, int32_t* argmin_out_dev, void* stream) {
  if (B <= 0 || N <= 0 || K <= 0 || !pos_dev || !ref_dev || !idx_ref_dev || !idx_pos_dev || !rmsd_out_dev) return fail(CBD_ERR_ARG, "bad argument");
  HIPCHK(launch_symm_rmsd(B, N, K, pos_dev, ref_dev, idx_ref_dev, idx_pos_dev, rmsd_out_dev, argmin_out_dev, reinterpret_cast<hipStream_t>(stream)));
  return 0;
}

}  // extern "C"
