/* cbdock.h -- C ABI of the MI355X (gfx950) reverse-diffusion docking engine.
 *
 * Drop-in boundary for ONE hot path of LDeng0205/confidence-bootstrapping: the score-model forward
 * pass and the reverse-SDE pose update that `utils/sampling.sampling()` drives.  The reference has no
 * FFI layer (it is pure Python over PyTorch/e3nn/PyG), so each entry point cites the Python interface it
 * stands behind; the Python host shim (confidence_bootstrapping_amd/engine.py, ctypes) is the only
 * intended caller, and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only, no torch types.  Every function returns 0 on success and a
 * negative cbd_status otherwise (cbd_last_error() gives the text); the Python shim raises RuntimeError so
 * the reference's "catch, halve batch_size, retry" protocol (inference.py:566-570) keeps working.
 * Caller owns every buffer passed in; weights/complex data are COPIED.  Pointers named *_host are host
 * memory; pointers named *_dev are ROCm device memory on the engine's device.  One engine per GPU,
 * not thread-safe, independent across engines.  All floating point is IEEE fp32, indices int32/int64 as typed.
 */
#ifndef CBDOCK_H
#define CBDOCK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cbd_engine cbd_engine;

typedef enum {
  CBD_OK = 0,
  CBD_ERR_ARG = -1,        /* bad argument / unsupported configuration            */
  CBD_ERR_HIP = -2,        /* a HIP runtime call failed                            */
  CBD_ERR_STATE = -3,      /* call order violated (weights/complex not set yet)    */
  CBD_ERR_CAPACITY = -4,   /* batch larger than the capacity given to cbd_create  */
  CBD_ERR_WEIGHT = -5      /* unknown tensor name or shape mismatch                */
} cbd_status;

/* Architecture + schedule constants.  Mirrors the keys of workdir/pretrained_score/model_parameters.yml that
 * utils/utils.py:239-283 maps onto TensorProductScoreModel.__init__ (models/score_model.py:45-56).  Only the
 * shipped architecture is implemented: ns=32, nv=6, sh_lmax=1, 3 embedding + 5 interaction layers. */
typedef struct {
  int32_t ns, nv;                   /* 32, 6                                              */
  int32_t num_conv_layers;          /* 5                                                  */
  int32_t num_prot_emb_layers;      /* 3                                                  */
  int32_t lm_embedding_dim;         /* 1280 ('precomputed') or 0                          */
  int32_t no_torsion;               /* 0/1                                                */
  float lig_max_radius;             /* 5    (args.max_radius)                             */
  float rec_max_radius;             /* 30                                                 */
  float cross_max_distance;         /* 80                                                 */
  float center_max_distance;        /* 30                                                 */
  int32_t lig_radius_cap;           /* 32   torch_cluster max_num_neighbors default       */
  int32_t max_batch;                /* capacity: poses per cbd_score / cbd_sample call    */
  int32_t device;                   /* HIP device ordinal                                 */
} cbd_config;

/* Per-step scalars.  The reference computes these on the HOST in numpy/torch scalar code every step
 * (utils/sampling.py:94-141, models/score_model.py:338,347,419-420,447 via utils/so3.py:90-94 and
 * utils/torus.py:78-82); the Python shim does the same and hands them over, so the look-up tables and
 * float64 scalar arithmetic stay bit-identical to the reference. */
typedef struct {
  float t;                 /* diffusion time (tr = rot = tor schedules coincide on this path)            */
  float tr_sigma;          /* t_to_sigma on fp32 tensors, score_model.py:338                             */
  float cross_cutoff;      /* 3*tr_sigma + 20, score_model.py:347                                        */
  float rot_score_norm;    /* so3.score_norm(rot_sigma), score_model.py:420                              */
  float tor_score_norm_sqrt; /* sqrt(torus.score_norm(tor_sigma)), score_model.py:447                    */
  float tr_score_coef, tr_noise_coef;    /* g^2 dt (or 0.5 g^2 dt for ODE) and g sqrt(dt), sampling.py:119-132 */
  float rot_score_coef, rot_noise_coef;
  float tor_score_coef, tor_noise_coef;
  float sigma_emb[32];     /* sinusoidal_embedding(embedding_scale * t_tr, 32), diffusion_utils.py:99-110: what the RECEPTOR side
                            * embeds (complex_t['tr'], score_model.py:323)                                */
  float sigma_emb_t[32];   /* the embedding the LIGAND nodes / edges, the cross and centre edges and the tr / rot magnitude heads use:
                            * equal to sigma_emb, except for a model built with asyncronous_noise_schedule, where it is the embedding
                            * of the common time t (node_t['t'] / complex_t['t'], score_model.py:408,460,497;
                            * utils/diffusion_utils.py:172-175)                                           */
} cbd_step;

const char* cbd_last_error(void);
const char* cbd_version(void);

/* TensorProductScoreModel.__init__ + .to(device)  (models/score_model.py:44-280, utils/utils.py:285-287). */
int cbd_create(const cbd_config* cfg, cbd_engine** out);
int cbd_destroy(cbd_engine* e);

/* model.load_state_dict (inference.py:307): one call per state-dict tensor, name = checkpoint key
 * (e.g. "conv_layers.0.fc.2.3.weight"), fp32 host data, row-major, shape as in the checkpoint.
 * Unknown keys under final_conv.tp. / tor_bond_conv.tp. / final_tp_tor. are accepted and ignored. */
int cbd_load_weight(cbd_engine* e, const char* name, const float* data_host, const int64_t* shape, int32_t ndim);
/* Re-pack the weights into the MFMA operand streams; fails if a required tensor was never loaded (strict=True). */
int cbd_finalize_weights(cbd_engine* e);

/* The complex that the batch consists of copies of (utils/sampling.py assumes every sample in a batch is the
 * same molecule, diffusion_utils.py:62-63).  Graph schema = datasets/process_mols.py:448-489,567-589:
 *   lig_x [Nl,16] categorical features; lig_bond_index [2, n_bond_dir] (each bond twice, consecutive);
 *   lig_bond_attr [n_bond_dir,4]; edge_mask [n_bond_dir] (0/1); mask_rotate [R, Nl] (0/1);
 *   rec_x [Nr, 1+lm_dim] (col 0 residue type); rec_pos [Nr,3]; rec_edge_index [2, Err].
 * Also runs the time-independent receptor embedding once (score_model.py:297-320). */
int cbd_set_complex(cbd_engine* e, int32_t Nl, int32_t Nr, int32_t n_bond_dir, int32_t R, int32_t Err,
                    const int64_t* lig_x_host, const int64_t* lig_bond_index_host, const float* lig_bond_attr_host,
                    const uint8_t* edge_mask_host, const uint8_t* mask_rotate_host,
                    const float* rec_x_host, const float* rec_pos_host, const int64_t* rec_edge_index_host);

/* model(batch) -- TensorProductScoreModel.forward (models/score_model.py:333-449) for B copies of the complex
 * at ligand poses pos [B,Nl,3].  Outputs: tr [B,3], rot [B,3], tor [B*R].  Device pointers; runs on `stream`
 * (a hipStream_t passed as void*, NULL = default stream) and returns without synchronising. */
int cbd_score(cbd_engine* e, int32_t B, const float* pos_dev, const cbd_step* step_host,
              float* tr_dev, float* rot_dev, float* tor_dev, void* stream);

/* modify_conformer_batch (utils/diffusion_utils.py:60-78): rigid update + sequential torsions + Kabsch
 * re-alignment.  tor_dev may be NULL (rigid only).  pos updated in place ([B,Nl,3]). */
int cbd_modify_conformer(cbd_engine* e, int32_t B, float* pos_dev, const float* tr_dev, const float* rot_dev,
                         const float* tor_dev, void* stream);

/* The step loop of sampling() (utils/sampling.py:93-223) for one batch: S steps of score -> perturbation ->
 * pose update.  noise_* are the N(0,1) draws the reference takes from torch.normal, lifted into inputs:
 * noise_tr [S,B,3], noise_rot [S,B,3], noise_tor [S,B*R] (NULL => zeros, i.e. no_random).  A step whose
 * *_noise_coef is 0 ignores its noise (no_final_step_noise / ODE).  pos_dev [B,Nl,3] is updated in place.
 * If scores_out_dev != NULL it receives the per-step scores [S, B*(6+R)] (tr,rot,tor per step) for tests. */
int cbd_sample(cbd_engine* e, int32_t B, int32_t S, const cbd_step* steps_host, float* pos_dev,
               const float* noise_tr_dev, const float* noise_rot_dev, const float* noise_tor_dev,
               float* scores_out_dev, void* stream);

/* Two batches (normally the pose batches of two different complexes) advanced through the same S steps in lockstep: every
 * tensor-product launch covers the edge groups of BOTH engines (twice the waves per launch; the per-launch drain of the
 * long-lived waves is amortised).  e0 and e1 must live on the same device and share one set of weights
 * (cbd_share_weights).  Results are identical to two cbd_sample calls.  Not captured into a hipGraph. */
int cbd_sample_pair(cbd_engine* e0, cbd_engine* e1, int32_t B0, int32_t B1, int32_t S, const cbd_step* steps_host, float* pos0_dev,
                    const float* noise_tr0_dev, const float* noise_rot0_dev, const float* noise_tor0_dev, float* pos1_dev,
                    const float* noise_tr1_dev, const float* noise_rot1_dev, const float* noise_tor1_dev, void* stream);

/* The same for n = 1..8 engines (arrays of n entries; the noise arrays or their entries may be NULL). */
int cbd_sample_multi(int32_t n, cbd_engine* const* engines, const int32_t* B, int32_t S, const cbd_step* steps_host,
                     float* const* pos_dev, const float* const* noise_tr_dev, const float* const* noise_rot_dev,
                     const float* const* noise_tor_dev, void* stream);

/* Engine options.  "graph" (0/1): capture the S-step loop of cbd_sample into a hipGraph that is instantiated once per
 * (batch size, schedule) and replayed with one launch per batch (inputs are staged into engine-owned buffers).
 * "bf16" (0/1): run the two Linears of every tensor-product layer's radial MLP on bf16 matrix cores (bf16 operands, fp32
 * accumulate; everything else stays fp32) -- BASELINE.json configs[3]; results differ from the fp32 path at the 1e-2 level.
 * "f32_split" (0/1): the same two Linears with every fp32 operand held as the exact sum of three bf16 planes (hi+mid+lo) and
 * the six plane products down to 2^-24 relative issued on the bf16 matrix cores with fp32 accumulate: results agree with the
 * fp32 kernel to fp32 rounding level (scores 2e-7..2e-6 relative, 20-step trajectories within 1e-4 A) at ~1.5x its
 * throughput.  "bf16" and "f32_split" are mutually exclusive: switching one on replaces the other, switching one off only
 * clears itself.  Default: both off = v_mfma_f32_32x32x2_f32 (exact fp32 products).
 * "async_setup" (0/1, default 0): cbd_set_complex of this engine works on a set-up stream (picked from a small per-device pool by
 * probing for one whose hardware queue is not behind a running step-loop graph) and waits only for the cbd_sample / cbd_sample_multi launches that used
 * THIS engine, instead of synchronising the device and using the default stream: the caller can set the next complexes up on idle
 * engines while others run (sampling.py does, with two alternating sets of engines).  A launch through any other entry point
 * (cbd_score, cbd_modify_conformer, cbd_recompute_receptor) makes the next set-up synchronise the device as before.
 * With "bf16": "bf16_stationary" (0/1) -- the 74 -> 74 layers through persistent workgroups that keep a whole FCBlock in registers
 * (tp_conv_bf16s.hip; same bf16 products, message sums equal to fp32 rounding; deterministic; default 1, 0 selects the streaming
 * kernel tp_conv_bf16.hip); baked into captured graphs (changing it drops the graphs); co-scheduled engines must agree on it.
 * "bf16_roles" (the role-split experiment of rounds 4-5) is not part of this library since round 6: 0 is accepted, any other value is
 * refused with CBD_ERR_ARG -- the experiment lives in the diagnostic twin library (tools/diag_lib.py, experiments/). */
int cbd_set_option(cbd_engine* e, const char* name, int64_t value);

/* Make `dst` use the device-resident (re-packed) weights of `src` instead of a copy of its own: several engines on one
 * GPU (one per HIP stream) then stream the same L2-resident weight tiles.  `src` must outlive `dst`. */
int cbd_share_weights(cbd_engine* dst, cbd_engine* src);

/* Re-run the time-independent receptor embedding of the current complex (models/score_model.py:297-320, which the
 * reference executes in the first model call of every batch).  cbd_set_complex already does this once; the
 * benchmark calls it per complex so that this work stays inside the timed region. */
int cbd_recompute_receptor(cbd_engine* e, void* stream);

/* Work accounting since the last reset (device-side counters, synchronises): out[0] = ligand-graph edges summed
 * over forward passes (each of the 3 ligand embedding layers visits them once), out[1] = edge visits of the 5
 * interaction layers (ALGORITHMIC: every sample's receptor->receptor edges of layer 0 are credited), out[2] = forward passes,
 * out[3] = the part of out[1] that is credited but not executed (layer-0 receptor->receptor messages depend on the diffusion time
 * only and are computed once per complex and shared by its B samples: (B - 1) * Err per forward pass). */
int cbd_stats(cbd_engine* e, int32_t reset, uint64_t out[4]);

/* Introspection used by the parity tests and the benchmark.  After a cbd_score call, copies a named
 * intermediate (see csrc/engine.hip: debug_names) to host memory; returns the element count or <0. */
int64_t cbd_debug_fetch(cbd_engine* e, const char* name, float* out_host, int64_t capacity);
/* Device-side edge counts of the last forward pass: [ll, lr, rr, rl, tor]. */
int cbd_last_edge_counts(cbd_engine* e, int64_t counts_host[5]);
/* Average duration (ms) of the dominant kernel (tp_conv) over the launches since the last reset, measured
 * with HIP events on the launch stream when timing is enabled; n_launches out.  enable: 0/1.  With "graph" = 1
 * the event pairs are event-record nodes of the captured graph (enable timing BEFORE the call that captures);
 * a graph's pairs are read before it is replayed again, so every replay is counted. */
int cbd_kernel_timing(cbd_engine* e, int32_t enable, int32_t reset, double* avg_ms_out, int64_t* n_launches_out,
                      double* total_ms_out);

/* Host-only helpers (no GPU needed), used by the CPU tests that emulate the MFMA tile algorithm:
 * re-pack one FCBlock (Linear 96->96 [w1 96x96, b1 96], Linear 96->W [w2 Wx96, b2 W]; reference models/layers.py:8-15 as
 * instantiated at models/tensor_layers.py:187-191) into the weight-tile stream of the tp_conv kernel.
 * in_level 0..3 / out_level 1..3 index get_irrep_seq (models/tensor_layers.py:21-26). */
int64_t cbd_conv_stream_floats(int32_t in_level, int32_t out_level);
int cbd_pack_conv_stream(int32_t in_level, int32_t out_level, const float* w1_host, const float* b1_host,
                         const float* w2_host, const float* b2_host, float* out_host);
/* The same for the layout the INFERENCE kernel (tp_conv.hip) reads: where the partly filled last tiles of the 1e and 0o blocks fit one
 * tile they share it (one tile less for the 2 -> 3 and 3 -> 3 layers).  cbd_pack_conv_stream is the layout of the training entry points
 * (cbd_tp_forward ...) and of the bf16 kernel. */
int64_t cbd_conv_stream_floats_infer(int32_t in_level, int32_t out_level);
int cbd_pack_conv_stream_infer(int32_t in_level, int32_t out_level, const float* w1_host, const float* b1_host, const float* w2_host,
                               const float* b2_host, float* out_host);


/* Symmetry-corrected ligand RMSD of B poses against one reference pose: replaces the loop over graph
 * isomorphisms of the reference's vendored spyrmsd (utils/molecules_utils.py:3-18 -> spyrmsd/rmsd.py:116-203, center=False,
 * minimize=False).  idx_ref / idx_pos [K][N] int32: isomorphism k maps reference atom idx_ref[k][i] to pose atom idx_pos[k][i]
 * (enumerated on the host).  Outputs: rmsd [B] and (optional) the index of the minimising isomorphism.  Device pointers. */
int cbd_symm_rmsd(int32_t B, int32_t N, int32_t K, const float* pos_dev, const float* ref_dev, const int32_t* idx_ref_dev,
                  const int32_t* idx_pos_dev, float* rmsd_out_dev, int32_t* argmin_out_dev, void* stream);

/* Neighbour graphs of the receptor featurisation (reference datasets/process_mols.py:456-479,491-513).
 * cbd_knn_graph replaces torch_cluster.knn_graph(pos, k) for one example (loop = False): nbr_out[i][r] = the r-th nearest node of
 * centre i, r < k <= n - 1, ties in distance to the lower index -- the caller forms edge_index = [nbr; centre].
 * cbd_radius_neighbors replaces the cdist / np.where / np.argsort loop of the non-kNN branch: for centre i the nodes closer than
 * `cutoff` in index order if at most `cap` (= max_neighbors), else the `cap` nearest by increasing distance; a centre with none gets
 * its nearest node; cnt_out[i] entries of idx_out[i][0..cap) are valid.  pos [n][3] fp32, device pointers. */
int cbd_knn_graph(int32_t n, int32_t k, const float* pos_dev, int32_t* nbr_out_dev, void* stream);
int cbd_radius_neighbors(int32_t n, float cutoff, int32_t cap, const float* pos_dev, int32_t* idx_out_dev, int32_t* cnt_out_dev,
                         void* stream);

/* ============================ all-atom CONFIDENCE model ===========================================
 * Replaces, for the shipped workdir/pretrained_confidence architecture, the confidence branch of
 * utils/sampling.py:240-261: crop_beyond (utils/utils.py:395-420) + set_time(0) + the all-atom
 * TensorProductScoreModel.forward in confidence mode (models/all_atom_score_model.py:363-454).
 * One engine per GPU and model; weights are loaded by state-dict name like cbd_load_weight. */
typedef struct cbd_conf_engine cbd_conf_engine;

typedef struct cbd_conf_config {
  int32_t ns, nv;                   /* 24, 6 (the only supported values)                       */
  int32_t num_conv_layers;          /* 5                                                       */
  int32_t lm_embedding_dim;         /* 1280 ('precomputed' ESM block) or 0                     */
  float lig_max_radius;             /* 5   : ligand radius graph, lig->atom edges              */
  float cross_cutoff;               /* 20  : lig->residue edges at t = 0 (dynamic_max_cross)   */
  int32_t lig_radius_cap;           /* 32  : torch_cluster max_num_neighbors                   */
  int32_t max_batch;                /* capacity: poses per cbd_conf_score call                 */
  int32_t device;                   /* HIP device ordinal                                      */
} cbd_conf_config;

int cbd_conf_create(const cbd_conf_config* cfg, cbd_conf_engine** out);
int cbd_conf_destroy(cbd_conf_engine* e);
/* One call per state_dict tensor of the confidence model (fp32, host), then finalize. */
int cbd_conf_load_weight(cbd_conf_engine* e, const char* name, const float* data_host, const int64_t* shape, int32_t ndim);
int cbd_conf_finalize_weights(cbd_conf_engine* e);
/* The un-cropped complex in the all-atom schema (datasets/process_mols.py:448-526), host pointers, copied.
 * lig_x [Nl][16] / atom_x [Na][4] categorical features as floats; rec_x [Nr][1 + lm_embedding_dim];
 * edge lists are [2][E] int64 rows (edge_index[0] = aggregating node); atom_res [Na] = residue of each atom. */
int cbd_conf_set_complex(cbd_conf_engine* e, int32_t Nl, int32_t Nr, int32_t Na, int32_t n_bond_dir, int32_t Err, int32_t Eaa,
                         const float* lig_x, const int64_t* bond_index, const float* bond_attr,
                         const float* rec_x, const float* rec_pos, const int64_t* rec_edge_index,
                         const float* atom_x, const float* atom_pos, const int64_t* atom_edge_index, const int64_t* atom_res);
/* Confidence of B poses of the current complex.  pos_dev [B][Nl][3] device fp32.  crop_beyond <= 0 disables the crop.
 * Outputs (device fp32): confidence_dev [B], atom_confidence_dev [B*Nl] (may be NULL).  Asynchronous on `stream`. */
int cbd_conf_score(cbd_conf_engine* e, int32_t B, const float* pos_dev, float crop_beyond, float* confidence_dev,
                   float* atom_confidence_dev, void* stream);
/* The same for the pose batches of up to four complexes (one engine each, all on one device) in ONE set of fused-conv launches: the
 * final poses of a co-scheduled group (cbd_sample_multi) are scored together, so a launch carries ~4x the waves and its last, partly
 * filled round of resident waves costs ~1 % instead of ~6 % (no reference counterpart: the reference scores one batch at a time,
 * utils/sampling.py:240-261).  Arrays of n entries; atom_confidence_dev (or any entry of it) may be NULL.  Results are bitwise those
 * of n cbd_conf_score calls.  Kernel timing (cbd_conf_kernel_timing) of the merged launches is recorded by engines[0]. */
int cbd_conf_score_multi(int32_t n, cbd_conf_engine* const* engines, const int32_t* B, const float* const* pos_dev, float crop_beyond,
                         float* const* confidence_dev, float* const* atom_confidence_dev, void* stream);
/* After the work of the last cbd_conf_score has completed: 0 if EVERY cbd_conf_score since the previous check was valid,
 * CBD_ERR_CAPACITY if a per-atom edge capacity was exceeded in any of them (their results must be discarded; the flag is sticky
 * and cleared by this call, so several batches -- also of different complexes -- can be scored back to back and checked once).
 * Synchronises the stream of the last call. */
int cbd_conf_check(cbd_conf_engine* e);
/* Options: "debug" (0/1) keeps per-layer ligand features of the next calls for cbd_conf_debug_fetch (synchronises). */
int cbd_conf_set_option(cbd_conf_engine* e, const char* name, int64_t value);
/* Introspection for parity tests / benchmark: named intermediate of the last call (see cbd_debug_fetch);
 * edge counts of the 9 groups (ll, lr, la, rr, rl, ra, aa, al, ar); HIP-event timing of the fused conv kernel. */
int64_t cbd_conf_debug_fetch(cbd_conf_engine* e, const char* name, float* out_host, int64_t capacity);
int cbd_conf_last_edge_counts(cbd_conf_engine* e, int64_t counts_host[9]);
int cbd_conf_kernel_timing(cbd_conf_engine* e, int32_t enable, int32_t reset, double* avg_ms_out, int64_t* n_launches_out,
                           double* total_ms_out);
/* Weight-stream packer of one FCBlock of a confidence conv layer (exposed for the CPU pack-emulation test). */
int64_t cbd_conf_stream_floats(int32_t in_level, int32_t out_level);
int cbd_conf_pack_stream(int32_t in_level, int32_t out_level, const float* w1_host, const float* b1_host,
                         const float* w2_host, const float* b2_host, float* out_host);

/* ---- training-side tensor product (fine-tuning step) ------------------------------------------------
 * Forward / backward of  msg[e] = FasterTensorProduct(x[e], [1, sqrt3 vec[e]], W2 h[e] + b2)
 * (reference models/tensor_layers.py:66-117 fed by the last Linear of the FCBlock, models/layers.py:8-15, inside
 * TensorProductConvLayer.forward models/tensor_layers.py:195-206), the op autograd differentiates in
 * utils/training.py:184-233 (train_epoch: loss.backward()).  All pointers are device pointers, all tensors fp32 and per edge:
 *   xrow [E][80]  features of the node the edge reads (node_attr[edge_dst]; columns >= the level's width are zero)
 *   vec4 [E][4]   unit edge vector (x, y, z, 0)
 *   h    [E][96]  hidden activations of the radial MLP (after ReLU and Dropout)
 *   n_groups, group_edges_host[n_groups], wstreams_dev[n_groups] (host arrays): the layer's edge groups (reference edge_groups /
 *                 fc[g], models/tensor_layers.py:190,201) -- consecutive edge ranges of the per-edge tensors, each with the tile
 *                 stream of its own FCBlock (cbd_pack_conv_stream layout, cbd_conv_stream_floats(in, out) floats); E = sum of
 *                 group_edges; 1 <= n_groups <= 4; all groups go out in ONE launch
 *   msg  [E][80]  out: messages (columns >= out width are zero)
 *   gmsg [E][80]  in:  d loss / d msg
 *   gx   [E][80]  out: d loss / d xrow
 *   gw   [E][Wp]  out: d loss / d (packed per-edge weights), Wp = cbd_tp_packed_width(in, out); column (tile-3)*32 + row of the
 *                 stream's second-Linear tiles.  The caller finishes the Linear's backward: g_h = gw W2p, dW2p = gw^T h.
 * (in_level, out_level) in {(0,1), (1,2), (2,3), (3,3)} = node widths 32/50/68/74 of get_irrep_seq (tensor_layers.py:12-27). */
int64_t cbd_tp_packed_width(int32_t in_level, int32_t out_level);
int cbd_tp_forward(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges_host, const float* xrow_dev,
                   const float* vec4_dev, const float* h_dev, const float* const* wstreams_dev, float* msg_dev, void* stream);
int cbd_tp_backward(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges_host, const float* xrow_dev,
                    const float* vec4_dev, const float* h_dev, const float* const* wstreams_dev, const float* gmsg_dev, float* gx_dev,
                    float* gw_dev, void* stream);

/* g_h = g_w W2p without reading g_w from memory: the gradient of the radial MLP's hidden activations (autograd of fc[3],
 * models/layers.py:8-15 under utils/training.py:205) as a second pass over the edges that re-forms g_w = g_msg (x) mid tile by tile in
 * registers and multiplies it with the TRANSPOSED weight tiles on the matrix cores.  wstreams_t_dev[g]: per second-Linear tile T,
 * fragment f = 16 kb + s, lane (i, hf'): W2p[32 T + (s & 3) + 8 (s >> 2) + 4 hf'][32 kb + i] at float ((f >> 2) * 64 + lane) * 4 + (f & 3)
 * of the tile's 3072 floats, plus one zero tile behind the last (train_ops.StreamHub builds it).  gh_dev: [E][96]. */
int cbd_tp_backward_gh(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges_host, const float* xrow_dev,
                       const float* vec4_dev, const float* const* wstreams_t_dev, const float* gmsg_dev, float* gh_dev, void* stream);

/* dW2p = g_w^T h and db2p = column sums of g_w for the edges [e_lo, e_hi) of one edge group, without g_w in memory (edges as the MFMA
 * k dimension, g_w re-formed per tile from g_msg and the mids; see csrc/tp_train.hip).  partial_dev: [n_chunks][wp * 96 + wp] fp32 with
 * wp = cbd_tp_packed_width(): chunk c holds the contribution of its share of the edges, dW2p rows first ([wp][96], the row order of
 * cbd_tp_backward's g_w columns), then db2p [wp]; the caller adds the chunks.  With this and cbd_tp_backward_gh, cbd_tp_backward may be
 * called with gw_dev = NULL. */
int cbd_tp_backward_dw(int32_t in_level, int32_t out_level, int64_t e_lo, int64_t e_hi, const float* xrow_dev, const float* vec4_dev,
                       const float* h_dev, const float* gmsg_dev, int32_t n_chunks, float* partial_dev, void* stream);
/* cbd_tp_backward_dw for ALL edge groups of a layer in one launch: group g owns the next group_edges[g] (> 0) edge rows and n_chunks[g]
 * consecutive partial rows of partial_dev [sum n_chunks][wp * 96 + wp]; add a group's rows in order (cbd_partial_reduce). */
int cbd_tp_backward_dw_groups(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges, const int32_t* n_chunks,
                              const float* xrow_dev, const float* vec4_dev, const float* h_dev, const float* gmsg_dev, float* partial_dev,
                              void* stream);
/* First stage of the FCBlock in the fine-tuning step (reference models/layers.py:8-15: Linear(96, 96) -> ReLU -> Dropout; per edge group in
 * TensorProductConvLayer.forward, models/tensor_layers.py:195-206, under model.train()), all edge groups of a layer in ONE launch each way.
 * Group g owns the next group_edges[g] (> 0) rows of x_dev / hid_dev [E][96]; weight_dev[g] = nn.Linear.weight [96 out][96 in], bias_dev[g]
 * [96] (host arrays of device pointers).  hid = dropout_p(relu(x W_g^T + b_g)); the dropout mask is a counter-based hash of (seed_dev[0],
 * call, element index) -- seed_dev is DEVICE memory so that a hipGraph replay draws fresh masks from its input buffer; `call` tells the
 * layers of a step apart.  Backward: gpre = ghid / (1 - p) where hid > 0 (active and kept), else 0;  gx (may be NULL) = gpre W_g. */
int cbd_fc1_forward(int32_t n_groups, const int64_t* group_edges, const float* x_dev, const float* const* weight_dev, const float* const* bias_dev,
                    float p_drop, const int64_t* seed_dev, int64_t call, float* hid_dev, void* stream);
int cbd_fc1_backward(int32_t n_groups, const int64_t* group_edges, const float* ghid_dev, const float* hid_dev, const float* const* weight_dev,
                     float p_drop, float* gpre_dev, float* gx_dev, void* stream);
/* cbd_outer_accum for all edge groups of a layer in one launch: partial_dev [sum n_parts][cbd_outer_accum_part_floats()], group g's
 * rows consecutive. */
int cbd_outer_accum_groups(int32_t n_groups, const int64_t* group_edges, const int32_t* n_parts, const float* g_dev, const float* x_dev,
                           float* partial_dev, void* stream);
/* Fixed-order reduction of partial rows: segment s = the next seg_rows[s] rows of partial_dev [.][width]; the column sums of a segment
 * go to out_a_dev[s] (columns [0, split)) and out_b_dev[s] (columns [split, width)); host arrays of device pointers, 1..4 segments.
 * Rows are added in row order: bitwise repeatable, no atomics (stands in for the torch.sum calls behind the two weight-gradient passes). */
int cbd_partial_reduce(int32_t n_seg, const int32_t* seg_rows, int32_t width, int32_t split, const float* partial_dev, float* const* out_a_dev,
                       float* const* out_b_dev, void* stream);
/* Generic Linear layers of the fine-tuning step (every nn.Linear outside the FCBlocks' first stage: reference models/score_model.py:186-243),
 * forward and backward without a library GEMM.  x [n_rows][ldx >= in_dim], weight [out_dim][in_dim] (nn.Linear.weight), bias [out_dim] or NULL.
 * act 0: y = x W^T + b;  act 1: y = dropout_p(relu(x W^T + b)) with the hash mask of cbd_fc1_forward (seed_dev: device scalar; `call`).
 * Backward: gpre = gy (act 0) or gy / (1 - p) where y > 0 (act 1: y_dev = the forward output, gpre_dev [n_rows][out_dim] is written);
 * gx_dev (or NULL) [n_rows][in_dim] = gpre W;  partial_dev [cbd_linear_backward_chunks(n_rows)][out_dim * in_dim + out_dim] = per-chunk
 * dW | db -- add the chunks in order with cbd_partial_reduce (split = out_dim * in_dim). */
int cbd_linear_forward(int64_t n_rows, int32_t in_dim, int32_t out_dim, const float* x_dev, int32_t ldx, const float* weight_dev,
                       const float* bias_dev, int32_t act, float p_drop, const int64_t* seed_dev, int64_t call, float* y_dev, void* stream);
int64_t cbd_linear_backward_chunks(int64_t n_rows);
int cbd_linear_backward(int64_t n_rows, int32_t in_dim, int32_t out_dim, const float* gy_dev, const float* y_dev, const float* x_dev, int32_t ldx,
                        const float* weight_dev, int32_t act, float p_drop, float* gpre_dev, float* gx_dev, float* partial_dev, void* stream);

/* Weight and bias gradient of the FCBlock's first Linear (96 -> 96) in the fine-tuning step (autograd of fc[0] in
 * models/layers.py:8-15 under utils/training.py:205): partial[p] = [ sum_e g[e][m] x[e][n] (96 x 96, row-major) | sum_e g[e][m] (96) ]
 * over the p-th of n_parts contiguous chunks of the E edges; the caller adds the n_parts blocks (cbd_outer_accum_part_floats()
 * floats each).  g_dev, x_dev: [E][96] fp32 device pointers. */
int64_t cbd_outer_accum_part_floats(void);
int cbd_outer_accum(int64_t E, const float* g_dev, const float* x_dev, int32_t n_parts, float* partial_dev, void* stream);

/* Deterministic segmented sum for the fine-tuning step: out[n][c] = sum over k in [rowptr[n], rowptr[n+1]) of vals[perm[k]][c], added in
 * index order.  Replaces the atomic scatters of the reference's training graph -- torch_scatter.scatter in
 * TensorProductConvLayer.forward (models/tensor_layers.py:206) and autograd's index_add for node_attr[edge_dst] -- so that a training
 * step is bitwise repeatable.  vals [E][width], perm [E] (edges grouped by target row: a stable argsort of the scatter index),
 * rowptr [n_rows + 1]; out [n_rows][width].  Device pointers, int64 indices (torch's index dtype). */
int cbd_segment_sum(int64_t n_rows, int32_t width, const float* vals_dev, const int64_t* perm_dev, const int64_t* rowptr_dev,
                    float* out_dev, void* stream);

/* The same with every row divided by max(rowptr[n+1] - rowptr[n], 1): torch_scatter.scatter(..., reduce='mean') of
 * TensorProductConvLayer.forward (models/tensor_layers.py:206). */
int cbd_segment_mean(int64_t n_rows, int32_t width, const float* vals_dev, const int64_t* perm_dev, const int64_t* rowptr_dev,
                     float* out_dev, void* stream);

/* Backward of cbd_segment_mean: out[e] = g[index[e]] / max(rowptr[index[e] + 1] - rowptr[index[e]], 1); width a multiple of 4. */
int cbd_segment_mean_backward(int64_t n_edges, int32_t width, const float* g_dev, const int64_t* index_dev, const int64_t* rowptr_dev,
                              float* out_dev, void* stream);

/* Train-mode e3nn.nn.BatchNorm (0.5.0: affine, normalization='component', reduce='mean') of the fine-tuning step, as the reference's
 * TensorProductConvLayer applies it (models/tensor_layers.py:191-193, 208-209) under model.train() (utils/training.py:186), fused with
 * the layer's residual  out + pad(node_attr)  (:211-213).  x [n][ldx >= dim] (columns >= dim ignored; their gx is zero), out [n][dim] fp32; fields [n_fields][3] int32 = {first column, components,
 * index among the 0e (scalar, even) fields or -1}; weight / running_var / save_* [n_fields]; bias / running_mean [#0e fields] (may be
 * NULL when there is none).  0e fields are centred and biased; every field is scaled by weight / sqrt(mean of its squared components
 * over rows and components + eps); running statistics are updated in place with `momentum`.  res_dev (or NULL): [n][res_dim], added to
 * the first res_dim output columns.  save_mean / save_inv feed the backward pass, which takes g [n][dim] and returns gx [n][ldx], gw [n_fields], gb [#0e]
 * (the residual's gradient is g[:, :res_dim] itself).  Sums in a fixed order: bitwise repeatable.
 * exclude4 (HOST pointer or NULL): {lo0, hi0, lo1, hi1}, two ascending disjoint row ranges left out of the statistics (the filler
 * graph's nodes of a capacity-padded, hipGraph-captured training step): their output rows are zero, and so is their input gradient. */
int cbd_irreps_bn_forward(int64_t n, int32_t dim, int32_t ldx, int32_t n_fields, const int32_t* fields_dev, const float* x_dev, const float* res_dev,
                          int32_t res_dim, const float* weight_dev, const float* bias_dev, float* running_mean_dev, float* running_var_dev,
                          float momentum, float eps, float* out_dev, float* save_mean_dev, float* save_inv_dev, const int64_t* exclude4,
                          void* stream);
int cbd_irreps_bn_backward(int64_t n, int32_t dim, int32_t ldx, int32_t n_fields, const int32_t* fields_dev, const float* g_dev, const float* x_dev,
                           const float* weight_dev, const float* save_mean_dev, const float* save_inv_dev, float* gx_dev, float* gw_dev,
                           float* gb_dev, const int64_t* exclude4, void* stream);
/* cbd_segment_sum over rows of stride ld >= width (only the first `width` columns are summed): the backward of cbd_gather_pad. */
int cbd_segment_sum_ld(int64_t n_rows, int32_t width, int32_t ld, const float* vals_dev, const int64_t* perm_dev, const int64_t* rowptr_dev,
                       float* out_dev, void* stream);

/* Edge-row assembly of the fine-tuning step, one launch each way.
 * cbd_edge_cat: out[e] = [edge_attr[e] (32) | node[src[e]][:32] | node[dst[e]][:32]] -- the FCBlock input of a layer, reference
 *   models/score_model.py:319,327,367.  node rows have stride node_ld (even, >= 32).  Backward: g [E][96] -> g_node [n_nodes][node_dim]
 *   = fixed-order sums of g[:, 32:64] over the edges grouped by src plus g[:, 64:96] over the edges grouped by dst (perm / rowptr from
 *   cbd_csr_build), columns >= 32 zero; the gradient of edge_attr is g[:, :32] itself.
 * cbd_gather_pad: out[e][:node_dim] = node[index[e]], out[e][node_dim:out_ld] = 0 -- `node_attr[edge_dst]` of
 *   models/tensor_layers.py:203 as the 80-float rows cbd_tp_forward reads; its backward is cbd_segment_sum_ld. */
int cbd_edge_cat(int64_t n_edges, const float* edge_attr_dev, const float* node_dev, int32_t node_ld, const int64_t* src_dev,
                 const int64_t* dst_dev, float* out_dev, void* stream);
int cbd_edge_cat_backward(int64_t n_nodes, int32_t node_dim, const float* g_dev, const int64_t* perm_src_dev, const int64_t* rowptr_src_dev,
                          const int64_t* perm_dst_dev, const int64_t* rowptr_dst_dev, float* g_node_dev, void* stream);
int cbd_gather_pad(int64_t n_edges, int32_t node_dim, int32_t out_ld, const float* node_dev, const int64_t* index_dev, float* out_dev,
                   void* stream);

/* Batched radius search of the fine-tuning step: torch_cluster.radius / radius_graph as the reference's training forward calls them
 * (models/score_model.py:498-503, 573-580, 652-656).  For every query y[q] (graph ybatch[q]) the points x[j], j in
 * [xptr[b], xptr[b+1]) of the same graph with |x/c - y/c|^2 < r2 (c = cutoff[b], or 1 when cutoff_dev is NULL), the first `cap` of them
 * in index order; with drop_self the point j == q counts towards the cap but is not returned (radius_graph: cap = max_neighbors + 1).
 * cbd_radius_count writes the number of edges per query; the caller scans them (exclusive) and passes the offsets to cbd_radius_fill,
 * which writes (query, point) pairs, sorted by query then point.  x, y: [n][3] fp32.  The distance arithmetic rounds like the torch
 * ops of the mask formulation (x - y, separate multiply and add, IEEE division), so the edge set is identical bit for bit. */
int cbd_radius_count(int64_t n_query, const float* x_dev, const float* y_dev, const float* cutoff_dev, float r2, const int64_t* xptr_dev,
                     const int64_t* ybatch_dev, int64_t cap, int32_t drop_self, int64_t* counts_dev, void* stream);
int cbd_radius_fill(int64_t n_query, const float* x_dev, const float* y_dev, const float* cutoff_dev, float r2, const int64_t* xptr_dev,
                    const int64_t* ybatch_dev, int64_t cap, int32_t drop_self, const int64_t* offsets_dev, int64_t* out_query_dev,
                    int64_t* out_point_dev, void* stream);

/* Edge geometry of the training forward in one launch: vec[e] = pos_b[idx_b[e]] - pos_a[idx_a[e]] (an index pointer may be NULL:
 * identity), raw4 = (vec, 0), unit4 = (vec / max(|vec|, 1e-12), 0) -- the `vec` operand of cbd_tp_forward --, and the Gaussian distance
 * expansion smear[e][k] = exp(coeff (|vec| - mu[k])^2) of models/score_model.py:667-677.  Any of the three outputs may be NULL. */
int cbd_edge_geometry(int64_t n_edges, const float* pos_a_dev, const float* pos_b_dev, const int64_t* idx_a_dev, const int64_t* idx_b_dev,
                      int32_t n_mu, const float* mu_dev, float coeff, float* raw4_dev, float* unit4_dev, float* smear_dev, void* stream);

/* The two e3nn heads of the score model in the fine-tuning step, one launch each way (csrc/train_heads.hip).
 * Centre convolution: final_conv.tp = o3.FullyConnectedTensorProduct(74-irreps, '1x0e+1x1o', '2x1o+2x1e') with per-edge weights
 * (reference models/score_model.py:245-255, 393-404 under utils/training.py:203-205): x [n][ldx] (74 columns read), vec [n][3] (not
 * normalised; no gradient), w [n][124] instruction-major, out [n][12] = [2x1o | 2x1e]; backward writes gx [n][ldx] and gw [n][124].
 * Torsion head: final_tp_tor (o3.FullTensorProduct('1x0e+1x1o', '2e')) + tor_bond_conv.tp (models/score_model.py:257-274, 431-441):
 * x [n][ldx], edge_vec / bond_vec [n][3], w [n][384], out [n][64] = [32x0o | 32x0e]. */
int cbd_center_tp_forward(int64_t n, const float* x_dev, int32_t ldx, const float* vec_dev, const float* w_dev, float* out_dev, void* stream);
int cbd_center_tp_backward(int64_t n, const float* x_dev, int32_t ldx, const float* vec_dev, const float* w_dev, const float* gout_dev,
                           float* gx_dev, float* gw_dev, void* stream);
int cbd_bond_tp_forward(int64_t n, const float* x_dev, int32_t ldx, const float* edge_vec_dev, const float* bond_vec_dev, const float* w_dev,
                        float* out_dev, void* stream);
int cbd_bond_tp_backward(int64_t n, const float* x_dev, int32_t ldx, const float* edge_vec_dev, const float* bond_vec_dev, const float* w_dev,
                         const float* gout_dev, float* gx_dev, float* gw_dev, void* stream);

/* Denoising score-matching loss of the fine-tuning step and its gradient with respect to the predictions in one launch
 * (reference utils/training.py:17-126 `loss_function` with apply_mean=True, differentiated by `loss.backward()` at utils/training.py:205).
 * tr / rot tensors [n_graphs][3], tr_sigma / rot_norm [n_graphs], torsion tensors [n_tor]; has_tor = 0 for no_torsion models.
 * out11 = the reference's 11-tuple (loss, tr, rot, tor, 0, 0, tr_base, rot_base, tor_base, 0, 0); g_* = d loss / d prediction.
 * Sums in double in a fixed order.  A batch without rotatable bonds gives NaN torsion terms, like the mean of an empty tensor. */
int cbd_score_loss(int32_t n_graphs, int32_t n_tor, int32_t has_tor, const float* tr_pred_dev, const float* tr_score_dev,
                   const float* tr_sigma_dev, const float* rot_pred_dev, const float* rot_score_dev, const float* rot_norm_dev,
                   const float* tor_pred_dev, const float* tor_score_dev, const float* tor_norm2_dev, float tr_weight, float rot_weight,
                   float tor_weight, float* out11_dev, float* g_tr_dev, float* g_rot_dev, float* g_tor_dev, void* stream);

/* Edge grouping for cbd_segment_sum without a host synchronisation: perm[n] = STABLE argsort of index[n] (values in [0, n_rows)),
 * rowptr[r] = number of indices < r for r in [0, n_rows] (what `torch.argsort(index, stable=True)` + a bincount/cumsum give the training
 * graph of utils/training.py:198-205; torch's stable sort synchronises the stream).  Everything is enqueued on `stream`; scratch is the
 * caller's: call once with scratch_dev = NULL to get *scratch_needed, then with a device buffer of at least that size. */
int cbd_csr_build(int64_t n, int64_t n_rows, const int64_t* index_dev, int64_t* perm_dev, int64_t* rowptr_dev, void* scratch_dev,
                  size_t scratch_bytes, size_t* scratch_needed, void* stream);

/* cbd_csr_build for up to 32 index tensors in ONE sort (the 18 groupings of a training step: 108 launches one by one, 7 this way):
 * segment s contributes keys (s << bits) | index_s[k].  index_dev / seg_n / seg_rows are HOST arrays (device pointers, element counts,
 * row counts); perm_dev receives the segments' permutations back to back (sum of seg_n), rowptr_dev their row pointers back to back
 * (seg_rows[s] + 1 each).  Sizing call and scratch as in cbd_csr_build. */
int cbd_csr_build_batched(int32_t n_seg, const int64_t* const* index_dev, const int64_t* seg_n, const int64_t* seg_rows, int64_t* perm_dev,
                          int64_t* rowptr_dev, void* scratch_dev, size_t scratch_bytes, size_t* scratch_needed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CBDOCK_H */
