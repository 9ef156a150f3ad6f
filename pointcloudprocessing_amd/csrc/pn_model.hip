// Whole-model plan: sequences the kernels of one PointNet forward / backward on a stream.
//
// Mirrors PointNet.call (point_cloud_analysis/pointnet/PointNet.py:197-292), TNet.call (:418-454) and the
// loss assembly of pointnet_train.py:334-351.  All host logic here is pointer arithmetic over a caller-owned
// workspace; nothing allocates, synchronises or keeps state, so a call can be captured into a hipGraph.
//
// Dataflow decisions (DESIGN.md has the long form):
//   * every ConvLayer stores only its pre-BN output z; BatchNormalization + ReLU are applied by the consumer on
//     load ("lazy operand"), statistics come from the producer's epilogue -> no normalisation / activation pass;
//   * the three 128->1024 layers never store their (B,N,1024) output: the epilogue keeps per-(cloud,channel)
//     max / argmax and the BN sums; their backward uses the Gram-matrix form (pn_maxbwd.hip);
//   * tf.matmul(pc, R) is folded into mlp_1_1's kernel (per-cloud 3x64 weights); tf.matmul(X, R_64) is a
//     per-cloud-weight contraction on the MFMA engine;
//   * tile+concat in front of seg_l1 is never formed: seg_l1's kernel is split into its 64 per-point rows and
//     its 1024 global rows, the latter applied once per cloud and added as a per-cloud bias.
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <string>
#include <vector>
#include "pn_internal.h"

namespace pn {

enum { BLK_IT = 0, BLK_M11, BLK_M12, BLK_FT, BLK_M21, BLK_M22, BLK_M23, BLK_C1, BLK_C2, BLK_C3, BLK_S1, BLK_S2, BLK_S3, BLK_S4,
       BLK_S5, N_BLOCKS };

struct Slot {
  std::string name;
  long long off;
  int rows, cols, kind, block;   // kind: 0 kernel 1 gamma 2 beta 3 moving_mean 4 moving_var 5 bias 6 tnet_w 7 tnet_b
};
struct LRef {  // offsets (floats) into the flat parameter buffer; -1 when absent
  long long kernel = -1, gamma = -1, beta = -1, mm = -1, mv = -1, bias = -1;
  int cin = 0, cout = 0, block = 0;
  bool has_bn = false;
};
struct TRef {
  LRef c1, c2, c3, d1, d2;
  long long w = -1, b = -1;
  int K = 0;
};
struct Layout {
  std::vector<Slot> slots;
  TRef iT, fT;
  LRef m11, m12, m21, m22, m23, c1, c2, c3, s1, s2, s3, s4, s5;
  long long total = 0;
};

static long long add_slot(Layout& L, const std::string& name, int rows, int cols, int kind, int block) {
  Slot s;
  s.name = name; s.rows = rows; s.cols = cols; s.kind = kind; s.block = block;
  s.off = L.total;
  L.total += ((long long)rows * cols + 63) / 64 * 64;   // 256-byte aligned slots
  L.slots.push_back(s);
  return s.off;
}
static LRef add_layer(Layout& L, const std::string& prefix, int cin, int cout, bool has_bn, int block) {
  LRef r;
  r.cin = cin; r.cout = cout; r.block = block; r.has_bn = has_bn;
  r.kernel = add_slot(L, prefix + ".kernel", cin, cout, 0, block);
  if (has_bn) {
    r.gamma = add_slot(L, prefix + ".bn.gamma", 1, cout, 1, block);
    r.beta = add_slot(L, prefix + ".bn.beta", 1, cout, 2, block);
    r.mm = add_slot(L, prefix + ".bn.moving_mean", 1, cout, 3, block);
    r.mv = add_slot(L, prefix + ".bn.moving_var", 1, cout, 4, block);
  } else {
    r.bias = add_slot(L, prefix + ".bias", 1, cout, 5, block);
  }
  return r;
}
static TRef add_tnet(Layout& L, const std::string& name, int K, int block) {
  TRef t;
  t.K = K;
  t.c1 = add_layer(L, name + ".conv1", K, 64, true, block);
  t.c2 = add_layer(L, name + ".conv2", 64, 128, true, block);
  t.c3 = add_layer(L, name + ".conv3", 128, 1024, true, block);
  t.d1 = add_layer(L, name + ".dense1", 1024, 512, true, block);
  t.d2 = add_layer(L, name + ".dense2", 512, 256, true, block);
  t.w = add_slot(L, name + ".w", 256, K * K, 6, block);
  t.b = add_slot(L, name + ".b", K, K, 7, block);
  return t;
}

static Layout make_layout(const pn_model_desc& d) {
  Layout L;
  if (!d.vanilla) L.iT = add_tnet(L, "input_transform", 3, BLK_IT);
  L.m11 = add_layer(L, "mlp_1_1", 3, 64, true, BLK_M11);
  L.m12 = add_layer(L, "mlp_1_2", 64, 64, true, BLK_M12);
  if (!d.vanilla) L.fT = add_tnet(L, "feature_transform", 64, BLK_FT);
  L.m21 = add_layer(L, "mlp_2_1", 64, 64, true, BLK_M21);
  L.m22 = add_layer(L, "mlp_2_2", 64, 128, true, BLK_M22);
  L.m23 = add_layer(L, "mlp_2_3", 128, 1024, true, BLK_M23);
  L.c1 = add_layer(L, "mlp_cls_1", 1024, 512, true, BLK_C1);
  L.c2 = add_layer(L, "mlp_cls_2", 512, 256, true, BLK_C2);
  L.c3 = add_layer(L, "mlp_cls_3", 256, d.ccls, false, BLK_C3);
  L.s1 = add_layer(L, "mlp_seg_1", 1088, 512, true, BLK_S1);
  L.s2 = add_layer(L, "mlp_seg_2", 512, 256, true, BLK_S2);
  L.s3 = add_layer(L, "mlp_seg_3", 256, 128, true, BLK_S3);
  L.s4 = add_layer(L, "mlp_seg_4", 128, 128, true, BLK_S4);
  L.s5 = add_layer(L, "mlp_seg_5", 128, d.cseg, false, BLK_S5);
  return L;
}

// ---- workspace ------------------------------------------------------------------------------------------
struct Arena {
  char* base = nullptr;
  size_t off = 0;
  size_t guard = ws_guard_bytes();   // debug: PN_WS_GUARD=<bytes> leaves an untouched gap after every entry (tools/ws_guard.py)
  std::vector<std::pair<std::string, std::pair<size_t, size_t>>>* dir = nullptr;
  static size_t ws_guard_bytes() {
    const char* e = getenv("PN_WS_GUARD");
    return e ? ((size_t)atoll(e) + 255) & ~(size_t)255 : 0;
  }
  template <class T>
  T* get(const char* name, size_t count) {
    const size_t bytes = ((count * sizeof(T) + 255) & ~(size_t)255) + guard;
    const size_t o = off;
    off += bytes;
    if (dir) dir->push_back({name, {o, count * sizeof(T)}});
    return base ? reinterpret_cast<T*>(base + o) : nullptr;
  }
};

struct CL {   // per-point conv layer state
  float *Z = nullptr, *part = nullptr, *mean = nullptr, *invstd = nullptr, *scale = nullptr, *shift = nullptr;
  float *ca = nullptr, *cb = nullptr, *cc = nullptr, *dy = nullptr;
  unsigned short *w16 = nullptr, *wt16 = nullptr;   // bf16 copies of the layer's (K, C) kernel, as it is / transposed (bf16 mode, K >= 64)
  int C = 0, K = 0;
};
struct ML {   // extra state of a max-pooled layer
  long long* pa1;
  float *g_all = nullptr, *zstar_all = nullptr;   // synchronised BatchNormalization: the pooled features / pre-BN maxima of ALL ranks' clouds
  float *pmax, *sumsq, *g, *zstar, *hs, *e, *nege, *f, *a1part, *a1, *gram, *GW, *Pm, *q, *D, *Wt, *We, *dG;
  float* gram_slabs = nullptr;     // the Gram job's slabs when the forward pass's finaliser carries it (gram_rides)
  int gram_rows = 0, gram_spc = 0;
  int *pq, *argq, *arg;            // per tile: 32-row block of the maximum; per cloud: the same after the reduction; the row (backward)
  unsigned short *wb_hi, *wb_lo;   // fragment-ordered bf16 copies of the kernel for the panel kernel (pn_panel.hip)
  int T64, tpc64, rows;   // panel tiles (all clouds / per cloud) and rows per panel
};
struct DLs {  // dense layer state (rows = B)
  float *z, *a, *mean, *invstd, *dz, *din;
  int K, C;
};
struct TN {
  CL c1, c2, c3;
  ML m3;
  DLs d1, d2;
  float *R, *dR, *da2;
};

struct WS {
  float *pcn, *cent, *scl;
  TN iT, fT;
  CL m11, m12, m21, m22, m23, s1, s2, s3, s4;
  ML mm23;
  DLs c1, c2, c3;
  float *Weff1, *dWeff1, *X64, *dX64, *tmpA12, *gb, *dgb, *dGseg, *dGcls;
  float *cls_logits, *cls_dlogits, *seg_dlogits, *seg_part, *dense_part, *slabs, *slabs_main, *bpart, *s5slab, *R3eye, *regpart;
  float* sync_part = nullptr;
  float* pm_slabs = nullptr;
  size_t slab_floats, slab_main_floats;
  float* slab_pool;          // slabs of the parameter-gradient jobs whose reduction is deferred to the end of the (phase of the) pass
  size_t slab_pool_floats;
  unsigned* dcount;
};

static long long wgrad_slab_rows(int B, int N, int Ci, int Cj, int* spc_out, int target_override = 0) {
  const int bm = (Ci % 128 == 0 && Cj % 128 == 0) ? 128 : 64, bn = (Cj % 128 == 0) ? 128 : 64;
  const int n_out = (Ci / bm) * (Cj / bn);
  // about one workgroup per CU: 512 slabs of 64 rows made the weight-gradient kernels write (and slab_reduce re-read) twice the
  // bytes for no extra parallelism -- 1.40 -> 1.34 ms/step at B=32, N=1024 (sweep: 64: 1.47, 128: 1.38, 192: 1.36, 256-384: 1.34)
  static const int target_blocks = getenv("PN_WGRAD_TARGET") ? atoi(getenv("PN_WGRAD_TARGET")) : 256;
  int target = (target_override > 0 ? target_override : target_blocks) / n_out;
  if (target < 1) target = 1;
  int spc = cdiv(target, B);
  const int max_spc = cdiv(N, 64);
  if (spc > max_spc) spc = max_spc;
  if (spc < 1) spc = 1;
  int rows = cdiv(cdiv(N, spc), 64) * 64;
  spc = cdiv(N, rows);
  *spc_out = spc;
  return rows;
}
static size_t wgrad_slab_floats(int B, int N, int Ci, int Cj) {
  int spc;
  wgrad_slab_rows(B, N, Ci, Cj, &spc);
  return (size_t)B * spc * Ci * Cj;
}

// a per-point tensor (M x C): fp32, or bf16 when the plan stores the layer-boundary tensors in 16 bits (PN_STORE_BF16); typed float*
// either way (every kernel that touches one takes the flag)
static float* plan_act(Arena& A, const std::string& name, size_t elems, bool s16) {
  if (s16) return reinterpret_cast<float*>(A.get<unsigned short>(name.c_str(), elems));
  return A.get<float>(name.c_str(), elems);
}
static void plan_cl(Arena& A, CL& l, const char* nm, long long M, int T, int C, bool store_z, bool training, bool s16, int K = 0) {
  std::string n(nm);
  l.C = C;
  l.K = K;
  if (s16 && K >= 64 && store_z) {       // the row GEMMs of this layer stage prepared bf16 copies of its kernel
    l.w16 = A.get<unsigned short>((n + ".w16").c_str(), (size_t)K * C);
    l.wt16 = A.get<unsigned short>((n + ".wt16").c_str(), (size_t)K * C);
  }
  l.Z = store_z ? plan_act(A, n + ".Z", (size_t)M * C, s16) : nullptr;
  l.part = A.get<float>((n + ".part").c_str(), (size_t)T * 2 * C);
  l.mean = A.get<float>((n + ".mean").c_str(), C);
  l.invstd = A.get<float>((n + ".invstd").c_str(), C);
  l.scale = A.get<float>((n + ".scale").c_str(), C);
  l.shift = A.get<float>((n + ".shift").c_str(), C);
  if (training) {
    l.ca = A.get<float>((n + ".ca").c_str(), C);
    l.cb = A.get<float>((n + ".cb").c_str(), C);
    l.cc = A.get<float>((n + ".cc").c_str(), C);
    l.dy = store_z ? plan_act(A, n + ".dy", (size_t)M * C, s16) : nullptr;
  }
}

static void plan_ml(Arena& A, ML& m, const char* nm, int B, int N, long long M, int T, int K, int C, bool training, int prec, int W = 1) {
  std::string n(nm);
  const int Bd = B * W;                            // rows of the per-cloud tensors that feed / come back from the dense layers
  const bool s16 = (prec & PN_STORE_BF16) != 0;
  m.rows = 64;
  m.tpc64 = panel_slots_per_cloud(B, N);   // slots (workgroups) per cloud of the panel kernel; never more than ceil(N / 64)
  m.T64 = B * m.tpc64;
  m.pmax = A.get<float>((n + ".pmax").c_str(), (size_t)B * cdiv(N, 64) * C);
  m.pq = A.get<int>((n + ".pq").c_str(), (size_t)B * cdiv(N, 64) * C);
  m.sumsq = A.get<float>((n + ".sumsq").c_str(), (size_t)B * cdiv(N, 64) * C);
  m.pa1 = nullptr;                                 // per cloud: fixed-point column sums of the staged rows -- carved from the block the step's first launch clears (plan_ws)
  m.argq = A.get<int>((n + ".argq").c_str(), (size_t)B * C);
  m.wb_hi = A.get<unsigned short>((n + ".wb_hi").c_str(), (size_t)K * C);
  m.wb_lo = A.get<unsigned short>((n + ".wb_lo").c_str(), (size_t)K * C);
  m.g = A.get<float>((n + ".g").c_str(), (size_t)B * C);
  m.zstar = A.get<float>((n + ".zstar").c_str(), (size_t)B * C);
  m.arg = A.get<int>((n + ".arg").c_str(), (size_t)B * C);
  if (W > 1) {
    m.g_all = A.get<float>((n + ".g_all").c_str(), (size_t)Bd * C);
    m.zstar_all = A.get<float>((n + ".zstar_all").c_str(), (size_t)Bd * C);
  }
  if (training) {
    m.hs = A.get<float>((n + ".hs").c_str(), (size_t)Bd * C);
    m.e = A.get<float>((n + ".e").c_str(), C);
    m.nege = A.get<float>((n + ".nege").c_str(), C);
    m.f = A.get<float>((n + ".f").c_str(), C);
    m.gram = A.get<float>((n + ".gram").c_str(), (size_t)K * K + K);   // A^T A followed by a1 = A^T 1 (one slab reduction for both)
    if (K == 128) {                  // slabs of the Gram job when it rides behind the forward pass's finaliser: 128 slabs in all
      m.gram_rows = (int)wgrad_slab_rows(B, N, K, K, &m.gram_spc, 128);
      m.gram_slabs = A.get<float>((n + ".gram_slabs").c_str(), (size_t)B * m.gram_spc * ((size_t)K * K + K));
    }
    m.a1 = m.gram + (size_t)K * K;
    m.a1part = nullptr;
    m.GW = A.get<float>((n + ".GW").c_str(), (size_t)K * C);
    m.Pm = A.get<float>((n + ".Pm").c_str(), (size_t)K * K);
    m.q = A.get<float>((n + ".q").c_str(), K);
    m.D = plan_act(A, n + ".D", (size_t)M * K, s16);
    m.Wt = A.get<float>((n + ".Wt").c_str(), (size_t)K * C);
    m.We = A.get<float>((n + ".We").c_str(), (size_t)K * C);
    m.dG = A.get<float>((n + ".dG").c_str(), (size_t)Bd * C);
  }
}
static void plan_dl(Arena& A, DLs& d, const char* nm, int B, int K, int C, bool training) {
  std::string n(nm);
  d.K = K; d.C = C;
  d.z = A.get<float>((n + ".z").c_str(), (size_t)B * C);
  d.a = A.get<float>((n + ".a").c_str(), (size_t)B * C);
  d.mean = A.get<float>((n + ".mean").c_str(), C);
  d.invstd = A.get<float>((n + ".invstd").c_str(), C);
  if (training) {
    d.dz = A.get<float>((n + ".dz").c_str(), (size_t)B * C);
    d.din = A.get<float>((n + ".din").c_str(), (size_t)B * K);
  }
}
static void plan_tn(Arena& A, TN& t, const char* nm, int B, int N, long long M, int T, int K, bool training, int prec, int W = 1) {
  std::string n(nm);
  const int Bd = B * W;
  const bool s16 = (prec & PN_STORE_BF16) != 0;
  plan_cl(A, t.c1, (n + ".c1").c_str(), M, T, 64, true, training, s16, K == 64 ? 64 : 0);
  plan_cl(A, t.c2, (n + ".c2").c_str(), M, T, 128, true, training, s16, 64);
  plan_cl(A, t.c3, (n + ".c3").c_str(), M, 1, 1024, false, training, s16);            // statistics live in the panel buffers (plan_ml)
  plan_ml(A, t.m3, (n + ".m3").c_str(), B, N, M, T, 128, 1024, training, prec, W);
  plan_dl(A, t.d1, (n + ".d1").c_str(), Bd, 1024, 512, training);
  plan_dl(A, t.d2, (n + ".d2").c_str(), Bd, 512, 256, training);
  t.R = A.get<float>((n + ".R").c_str(), (size_t)Bd * K * K);
  if (training) {
    t.dR = A.get<float>((n + ".dR").c_str(), (size_t)Bd * K * K);
    t.da2 = A.get<float>((n + ".da2").c_str(), (size_t)Bd * 256);
  }
}

static void plan_ws(Arena& A, WS& w, const pn_model_desc& d, int B, int N, bool training) {
  const long long M = (long long)B * N;
  const int T = B * cdiv(N, 128);
  const int W = (training && d.sync_world > 1) ? d.sync_world : 1;     // synchronised BatchNormalization: the dense side holds every rank's rows
  const int Bd = B * W;
  const bool s16 = (d.prec & PN_STORE_BF16) != 0;
  w.pcn = A.get<float>("pcn", (size_t)M * 3);
  w.cent = A.get<float>("centroid", (size_t)B * 3);
  w.scl = A.get<float>("scale", B);
  if (!d.vanilla) {
    plan_tn(A, w.iT, "iT", B, N, M, T, 3, training, d.prec, W);
    plan_tn(A, w.fT, "fT", B, N, M, T, 64, training, d.prec, W);
  }
  plan_cl(A, w.m11, "m11", M, T, 64, true, training, s16);
  plan_cl(A, w.m12, "m12", M, T, 64, true, training, s16, 64);
  plan_cl(A, w.m21, "m21", M, T, 64, true, training, s16, 64);
  plan_cl(A, w.m22, "m22", M, T, 128, true, training, s16, 64);
  plan_cl(A, w.m23, "m23", M, 1, 1024, false, training, s16);                            // statistics live in the panel buffers (plan_ml)
  plan_ml(A, w.mm23, "mm23", B, N, M, T, 128, 1024, training, d.prec, W);
  plan_cl(A, w.s1, "s1", M, T, 512, true, training, s16, 64);      // the 64 per-point rows of seg_l1's (1088, 512) kernel
  plan_cl(A, w.s2, "s2", M, T, 256, true, training, s16, 512);
  plan_cl(A, w.s3, "s3", M, T, 128, true, training, s16, 256);
  plan_cl(A, w.s4, "s4", M, T, 128, true, training, s16, 128);
  plan_dl(A, w.c1, "c1", Bd, 1024, 512, training);
  plan_dl(A, w.c2, "c2", Bd, 512, 256, training);
  plan_dl(A, w.c3, "c3", Bd, 256, d.ccls, training);
  w.Weff1 = A.get<float>("Weff1", (size_t)B * 3 * 64);
  w.X64 = d.vanilla ? nullptr : plan_act(A, "X64", (size_t)M * 64, s16);
  w.gb = A.get<float>("gb", (size_t)Bd * 512);
  w.cls_logits = A.get<float>("cls_logits", (size_t)Bd * d.ccls);
  w.pm_slabs = A.get<float>("pm_slabs", (size_t)32 * 128 * 128);      // the 32 prep workgroups' shares of a max-pooled layer's Pm (one layer at a time)
  w.sync_part = W > 1 ? A.get<float>("sync_part", (size_t)T * 2 * 512) : nullptr;      // all-reduced copy of a layer's per-tile partial sums
  // per 128-row block of seg_out_fwd, or per 64-row tile of the fused frozen head (never straddling clouds)
  w.seg_part = A.get<float>("seg_part", (size_t)std::max<long long>(cdivll(M, seg_out_part_rows()), (long long)B * cdiv(N, seg_head_fused_rows())) *
                                            seg_out_part_stride());
  w.dense_part = A.get<float>("dense_part", (size_t)8 * Bd * 4096);         // split-K tiles of the dense layers (<= 8 splits)
  // their in-launch arrival counters, followed by the three max-pooled layers' column-sum accumulators (B x 256 64-bit words each):
  // ONE block, cleared by the step's first launch
  w.dcount = A.get<unsigned>("dcount", DENSE_MAX_COUNTERS + 3 * (size_t)B * 512);
  {
    long long* acc = w.dcount ? reinterpret_cast<long long*>(w.dcount + DENSE_MAX_COUNTERS) : nullptr;
    if (!d.vanilla) {
      w.iT.m3.pa1 = acc;
      w.fT.m3.pa1 = acc ? acc + (size_t)B * 256 : nullptr;
    }
    w.mm23.pa1 = acc ? acc + 2 * (size_t)B * 256 : nullptr;
  }
  w.R3eye = A.get<float>("R3eye", (size_t)B * 9);
  w.regpart = A.get<float>("regpart", (size_t)2 * B);
  w.slab_floats = w.slab_main_floats = w.slab_pool_floats = 0;
  w.slabs = w.slabs_main = w.slab_pool = nullptr;
  if (training) {
    w.dWeff1 = A.get<float>("dWeff1", (size_t)B * 3 * 64);
    w.dX64 = plan_act(A, "dX64", (size_t)M * 64, s16);
    w.tmpA12 = plan_act(A, "tmpA12", (size_t)M * 64, s16);
    w.dgb = A.get<float>("dgb", (size_t)B * 512);
    w.dGseg = A.get<float>("dGseg", (size_t)Bd * 1024);
    w.dGcls = A.get<float>("dGcls", (size_t)Bd * 1024);
    w.cls_dlogits = A.get<float>("cls_dlogits", (size_t)Bd * d.ccls);
    w.seg_dlogits = A.get<float>("seg_dlogits", (size_t)M * d.cseg);
    w.bpart = A.get<float>("bpart", (size_t)T * 2 * 512);
    w.s5slab = A.get<float>("s5slab", (size_t)T * 128 * d.cseg);
    size_t sf = 0;
    const int shapes[][2] = {{64, 64}, {64, 128}, {128, 128}, {64, 512}, {512, 256}, {256, 128}};
    for (auto& s : shapes) {
      int spc_;
      wgrad_slab_rows(B, N, s[0], s[1], &spc_);
      const size_t f = wgrad_slab_floats(B, N, s[0], s[1]) + (size_t)B * spc_ * s[0];   // + the optional column-sum row per slab
      if (f > sf) sf = f;
    }
    const size_t c3f = (size_t)T * 3 * 64;
    if (c3f > sf) sf = c3f;
    const size_t pmf = wgrad_slab_floats(1, 1024, 128, 128);    // W diag(e) W^T of the max-pooled layers
    if (pmf > sf) sf = pmf;
    w.slab_floats = sf;
    w.slabs = A.get<float>("slabs", sf);
    // the weight-gradient-shaped jobs ON the data-gradient path (W diag(e) W^T, d(R_64), d(W_eff1)) get their own scratch, so the
    // parameter-gradient jobs can run beside them on the auxiliary stream
    size_t mf = pmf;
    if (c3f > mf) mf = c3f;
    const size_t rf = wgrad_slab_floats(B, N, 64, 64);
    if (rf > mf) mf = rf;
    w.slab_main_floats = mf;
    w.slabs_main = A.get<float>("slabs_main", mf);
    // one region per deferred job (Run::pool_take): every per-point layer whose kernel gradient nobody reads before the optimizer
    const int pool_shapes[][2] = {{64, 128}, {64, 64}, {64, 128}, {64, 64}, {64, 64}, {64, 128}, {64, 512}, {512, 256}, {256, 128}, {128, 128}};
    size_t pf = 2 * (c3f + 64);                            // the two Cin = 3 layers
    for (auto& s : pool_shapes) pf += wgrad_slab_floats(B, N, s[0], s[1]) + 64;
    {                                                      // + the three Gram matrices (A^T A and the column sums) of the max-pooled layers
      int spc_;
      wgrad_slab_rows(B, N, 128, 128, &spc_);
      pf += 3 * (wgrad_slab_floats(B, N, 128, 128) + (size_t)B * spc_ * 128 + 64);
    }
    w.slab_pool_floats = pf;
    w.slab_pool = A.get<float>("slab_pool", pf);
  }
}

// ---- the run context ------------------------------------------------------------------------------------
struct Run {
  const pn_model_desc& d;
  const pn_model_io& io;
  Layout L;
  WS w;
  int B, N, T, tpc, prec;      // prec: PN_PREC_* (| PN_STORE_BF16: what the per-point launchers are given)
  // Synchronised BatchNormalization (pn_model_desc.sync_world, training): W ranks, this one is rk; the per-cloud dense layers run on
  // the Bd = B * W rows of all ranks (gathered), every per-point statistic is summed over the ranks (pn_model_io.sync_hook)
  int W = 1, rk = 0, Bd = 0;
  template <class Tp> Tp* loc(Tp* all_rows, long long per_row) const { return all_rows ? all_rows + (long long)rk * B * per_row : nullptr; }
  int sync_call(int op, const void* src, void* dst, long long n, int dtype) {
    if (!io.sync_hook) { set_error("pn_model: sync_world > 1 needs pn_model_io.sync_hook"); return PN_ERR_INVALID_ARGUMENT; }
    if (io.sync_hook(io.sync_ctx, op, src, dst, n, dtype, st) != 0) { set_error("pn_model: sync_hook failed (op %d, %lld elements)", op, n); return PN_ERR_LAUNCH; }
    return PN_OK;
  }
  int sync_sum(const void* src, void* dst, long long n, int dtype = 0) { return W > 1 ? sync_call(0, src, dst, n, dtype) : (int)PN_OK; }
  // local rows (n_local elements at all_rows + rk * n_local) -> every rank's rows, in place
  int sync_gather_rows(float* all_rows, long long n_local) { return W > 1 ? sync_call(1, all_rows + (long long)rk * n_local, all_rows, n_local, 0) : (int)PN_OK; }
  int s16 = 0;                 // the per-point layer-boundary tensors (Z, dy, X64, dX64, D) are bf16
  long long M;
  hipStream_t st;
  bool training;
  float* P;
  float* G;
  // Parameter gradients are off the data-gradient chain: nothing reads them before the optimizer.  With an auxiliary
  // stream (pn_model_io.aux_stream) their launches are deferred and flushed to it at every layer boundary, behind an
  // event recorded on the main stream, so they overlap the chain; without one they run in place.  Under stream
  // capture the events become graph edges (two branches, joined at the end of pn_model_backward).
  hipStream_t aux = nullptr;
  std::vector<std::function<int()>> deferred;
  int n_flush = 0;
  bool on_aux = false;
  float* cur_slabs() const { return on_aux || !aux ? w.slabs : w.slabs_main; }
  // Deferred slab reductions (slab_reduce_batch): a job keeps its slabs in a region of its own until flush_jobs().  Not with an
  // auxiliary stream (its launches are already off the main chain), and PN_SLAB_DEFER=0 restores one reduction per layer.
  std::vector<SlabJob> jobs;
  std::vector<WgradDesc> wg_jobs;                    // their weight-gradient GEMMs, launched together by tile shape (conv_wgrad_batch)
  std::vector<WgradDesc> gw_jobs;                    // the G W products of the max-pooled layers (consume the reduced Gram matrices)
  std::vector<std::function<int()>> after_jobs;      // launches that consume a deferred reduction (and feed only the optimizer)
  std::vector<DwJob> dw_jobs;                        // ... and the dW kernels of the max-pooled layers behind those, one launch
  std::vector<DenseWgradJob> dense_jobs;             // the per-cloud dense layers' weight gradients (bwd_chain): one launch per pass
  int dw_K = 0, dw_C = 0;
  bool last_deferred = false;
  size_t pool_used = 0;
  float* pool_take(size_t floats) {
    static const bool on = !(getenv("PN_SLAB_DEFER") && atoi(getenv("PN_SLAB_DEFER")) == 0);
    floats = (floats + 63) & ~(size_t)63;
    if (!on || aux || !w.slab_pool || pool_used + floats > w.slab_pool_floats) return nullptr;
    float* r = w.slab_pool + pool_used;
    pool_used += floats;
    return r;
  }
  int flush_jobs() {
    int rc = wg_jobs.empty() ? PN_OK : conv_wgrad_batch(wg_jobs.data(), (int)wg_jobs.size(), st);
    wg_jobs.clear();
    if (rc == PN_OK && !jobs.empty()) rc = slab_reduce_batch(jobs.data(), (int)jobs.size(), st);
    jobs.clear();
    // the dense layers' weight gradients (independent of everything here) ride behind the G W products' 96 workgroups (PN_DENSE_RIDE=0: a
    // launch of their own at the end, as before)
    static const bool dense_ride = !(getenv("PN_DENSE_RIDE") && atoi(getenv("PN_DENSE_RIDE")) == 0);
    bool rode = false;
    if (rc == PN_OK && !gw_jobs.empty()) {
      if (dense_ride && !dense_jobs.empty() && dense_jobs.size() <= (size_t)DENSE_WGRAD_MAX_JOBS)
        rc = conv_wgrad_batch(gw_jobs.data(), (int)gw_jobs.size(), st, dense_jobs.data(), (int)dense_jobs.size(), &rode);
      else rc = conv_wgrad_batch(gw_jobs.data(), (int)gw_jobs.size(), st);
    }
    if (rode) dense_jobs.clear();
    gw_jobs.clear();
    for (auto& f : after_jobs) {
      if (rc != PN_OK) break;
      rc = f();
    }
    after_jobs.clear();
    if (rc == PN_OK && !dw_jobs.empty()) rc = maxbwd_dw_batch(dw_jobs.data(), (int)dw_jobs.size(), B, N, dw_K, dw_C, st);
    dw_jobs.clear();
    for (size_t q = 0; rc == PN_OK && q < dense_jobs.size(); q += DENSE_WGRAD_MAX_JOBS)
      rc = dense_wgrad_batch(dense_jobs.data() + q, (int)std::min<size_t>(DENSE_WGRAD_MAX_JOBS, dense_jobs.size() - q), st);
    dense_jobs.clear();
    return rc;
  }

  static hipEvent_t pooled_event(int i) {
    static thread_local std::vector<hipEvent_t> pool;
    while ((int)pool.size() <= i) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[i];
  }
  // run f on the auxiliary stream once the main stream reaches the next flush point (or right now without one)
  int side(std::function<int()> f) {
    if (!aux) return f();
    deferred.push_back(std::move(f));
    return PN_OK;
  }
  int n_sites = 0;
  int flush(bool force = false) {
    if (!aux || deferred.empty()) return PN_OK;
    static const int every = getenv("PN_AUX_FLUSH_EVERY") ? atoi(getenv("PN_AUX_FLUSH_EVERY")) : 1;
    if (!force && (++n_sites % every) != 0) return PN_OK;
    hipEvent_t e = pooled_event(n_flush++);
    if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(aux, e, 0) != hipSuccess) {
      set_error("pn_model_backward: fork to the auxiliary stream failed");
      return PN_ERR_LAUNCH;
    }
    const hipStream_t main_st = st;
    st = aux; on_aux = true;
    int rc = PN_OK;
    for (auto& f : deferred) {
      rc = f();
      if (rc != PN_OK) break;
    }
    st = main_st; on_aux = false;
    deferred.clear();
    return rc;
  }
  int join() {
    if (!aux) return PN_OK;
    PN_TRY(flush(true));
    if (n_flush == 0) return PN_OK;
    hipEvent_t e = pooled_event(n_flush++);
    if (!e || hipEventRecord(e, aux) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) {
      set_error("pn_model_backward: join of the auxiliary stream failed");
      return PN_ERR_LAUNCH;
    }
    return PN_OK;
  }

  bool tr(int block) const { return io.trainable ? io.trainable[block] != 0 : true; }
  // the segmentation head as one launch (pn_segout.hip: seg_head_fused): its BatchNormalization layers all use moving statistics, no
  // gradient will be asked through it, the bf16 kernel copies exist, and the caller does not want the layers' outputs kept
  bool fused_seg_head() const {
    if ((io.flags & PN_IO_KEEP_ACTIVATIONS) || !w.s1.wt16 || !w.s2.wt16 || !w.s3.wt16 || !w.s4.wt16) return false;
    if (bn_batch(BLK_S1) || bn_batch(BLK_S2) || bn_batch(BLK_S3) || bn_batch(BLK_S4)) return false;
    if (!training) return true;
    return io.labels_seg != nullptr && io.loss_weights[1] == 0.f;
  }
  bool bn_batch(int block) const { return training && tr(block); }
  float* p(long long off) const { return off >= 0 ? P + off : nullptr; }
  float* gr(long long off) const { return (G && off >= 0) ? G + off : nullptr; }

  pn_operand lazy(const CL& l) const {
    pn_operand o;
    memset(&o, 0, sizeof(o));
    o.s1 = l.Z; o.ca = l.scale; o.cc = l.shift; o.ld = l.C; o.lo = 0.f; o.h16 = s16;
    return o;
  }
  pn_operand plain_act(const float* x, long long ld) const {     // a per-point tensor as it is
    pn_operand o = plain(x, ld);
    o.h16 = s16;
    return o;
  }
  static pn_operand plain(const float* x, long long ld) {
    pn_operand o;
    memset(&o, 0, sizeof(o));
    o.s1 = x; o.ld = ld; o.lo = -INFINITY;
    return o;
  }
  pn_operand dzop(const CL& l) const {
    pn_operand o;
    memset(&o, 0, sizeof(o));
    o.s1 = l.dy; o.s2 = l.Z; o.ca = l.ca; o.cb = l.cb; o.cc = l.cc; o.ld = l.C; o.lo = -INFINITY; o.h16 = s16;
    return o;
  }

  // ---------------- forward pieces ----------------
  int bn_fin(const CL& l, const LRef& r, int n_tiles = -1) {
    const int ub = bn_batch(r.block) ? 1 : 0;
    if (!ub) return PN_OK;       // moving statistics: the coefficients were written by the pass's first launch (fwd_prologue)
    const int nt = n_tiles < 0 ? T : n_tiles;
    const float* part = l.part;
    if (W > 1) {                 // the tiles' sums of every rank, tile by tile (the rank's own partials stay: seg_l1's feed its backward)
      PN_TRY(sync_sum(l.part, w.sync_part, (long long)nt * 2 * r.cout));
      part = w.sync_part;
    }
    return bn_finalize(part, nt, r.cout, M * W, p(r.gamma), p(r.beta), p(r.mm), p(r.mv), d.bn_momentum, d.bn_eps, ub, ub, l.mean,
                       l.invstd, l.scale, l.shift, st);
  }
  int fwd_conv(CL& l, const LRef& r, const pn_operand& x, const float* W, long long wcs, const float* cloud_bias) {
    PN_TRY(conv_fwd(&x, W, wcs, B, N, r.cin, r.cout, cloud_bias, l.Z, bn_batch(r.block) ? l.part : nullptr, prec,
                    st, (wcs == 0 && W == p(r.kernel)) ? l.wt16 : nullptr));
    return bn_fin(l, r);
  }
  // Experiment (PN_GRAM_RIDE=1; off by default: measured SLOWER): the Gram matrix of a max-pooled layer's input (what its Gram-form
  // backward needs: forward activations only) rides behind the forward pass's finaliser instead of waiting for the backward pass's
  // tail.  A lone 256-workgroup weight-gradient job is a 12-15 us chain of dependent chunk loads, the finaliser it rides behind 7.5 us:
  // the launch grew to 15.1 us, +22.8 us per step against the 17.8 us tail launch it removes (C2 0.738 -> 0.749 ms).  Work hides behind
  // a launch only if its own chain is shorter than that launch's.  The same predicate in both passes.
  bool gram_rides(const ML& m, const LRef& r) const {
    static const bool on = getenv("PN_GRAM_RIDE") && atoi(getenv("PN_GRAM_RIDE")) == 1;
    return on && training && tr(r.block) && G && W == 1 && !aux && m.gram_slabs && r.cin == 128 && s16 && (prec & ~PN_STORE_BF16) == PN_PREC_BF16;
  }
  int fwd_max(CL& l, ML& m, const LRef& r, const pn_operand& x, int prof_slot) {
    const int ub = bn_batch(r.block) ? 1 : 0;
    void** ev = io.prof_events;
    if (ev && ev[2 * prof_slot]) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(ev[2 * prof_slot]), st);
    PN_TRY(conv_fwd_max_panel(&x, m.wb_hi, m.wb_lo, B, N, r.cin, r.cout, m.pmax, m.pq, ub ? m.sumsq : nullptr, ub ? m.pa1 : nullptr, prec, st));
    if (ev && ev[2 * prof_slot + 1]) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(ev[2 * prof_slot + 1]), st);
    if (W > 1 && ub) {           // slot by slot / (cloud, column) by (cloud, column): the finaliser's sums then run over every rank's rows
      PN_TRY(sync_sum(m.sumsq, m.sumsq, (long long)m.T64 * r.cout));
      PN_TRY(sync_sum(m.pa1, m.pa1, (long long)B * r.cin * ((prec & ~PN_STORE_BF16) == PN_PREC_BF16X3 ? 2 : 1), 1));
    }
    // one finaliser: the layer's BatchNormalization coefficients (+ moving statistics) and the reduce_max over each cloud's panels
    WgradDesc gram{x, x, B, N, r.cin, r.cin, m.gram_rows, m.gram_slabs, prec, 1, 0};
    PN_TRY(panel_finalize(m.pmax, m.pq, m.sumsq, m.pa1, m.wb_hi, m.wb_lo, prec, B, N, r.cin, r.cout, p(r.gamma), p(r.beta), p(r.mm), p(r.mv), d.bn_momentum, d.bn_eps, ub,
                          ub, l.mean, l.invstd, l.scale, l.shift, W > 1 ? loc(m.g_all, r.cout) : m.g, W > 1 ? loc(m.zstar_all, r.cout) : m.zstar, m.argq, st, W,
                          gram_rides(m, r) ? &gram : nullptr));
    if (W > 1) {                 // the dense layers behind the pool see every rank's clouds; the backward needs every rank's maxima
      PN_TRY(sync_gather_rows(m.g_all, (long long)B * r.cout));
      PN_TRY(sync_gather_rows(m.zstar_all, (long long)B * r.cout));
    }
    return PN_OK;
  }
  // the pooled features as the layers behind them read them: this rank's clouds, or (synchronised BN) the rows of all ranks
  const float* pooled(const ML& m) const { return W > 1 ? m.g_all : m.g; }
  const float* pooled_local(const ML& m, int C) const { return W > 1 ? loc(m.g_all, C) : m.g; }
  const float* zstar_local(const ML& m, int C) const { return W > 1 ? loc(m.zstar_all, C) : m.zstar; }
  // Inference (moving statistics everywhere): a max-pooled chain c1 -> c2 -> c3 + reduce_max as ONE launch (pn_panel.hip: chain_max_kernel),
  // bit-identical to the three launches it replaces.  bf16-storage mode only (it stages the prepared bf16 kernel copies and rounds
  // where that plan's stores round); not when the caller wants every layer's output kept (PN_IO_KEEP_ACTIVATIONS: check_numerics).
  bool fused_chain(const CL& c2) const {
    static const bool on = !(getenv("PN_CHAIN_FUSE") && atoi(getenv("PN_CHAIN_FUSE")) == 0);
    return on && !training && s16 && (prec & ~PN_STORE_BF16) == PN_PREC_BF16 && !(io.flags & PN_IO_KEEP_ACTIVATIONS) && c2.wt16 != nullptr;
  }
  // x == nullptr: the chain starts from the normalised cloud (first layer's (3, 64) kernel w1f); else from the 64-channel operand
  int fwd_chain(const pn_operand* x, const float* w1f, CL& c1, CL& c2, CL& c3, ML& m, const LRef& r3) {
    PN_TRY(chain_fwd_max(x, x ? nullptr : w.pcn, w1f, x ? c1.wt16 : nullptr, c1.scale, c1.shift, c2.wt16, c2.scale, c2.shift, m.wb_hi, B, N,
                         m.pmax, m.pq, st));
    return panel_finalize(m.pmax, m.pq, nullptr, nullptr, m.wb_hi, m.wb_lo, prec, B, N, r3.cin, r3.cout, p(r3.gamma), p(r3.beta), p(r3.mm), p(r3.mv),
                          d.bn_momentum, d.bn_eps, 0, 0, c3.mean, c3.invstd, c3.scale, c3.shift, m.g, m.zstar, m.argq, st);
  }
  // out (B, C) = x (B, K) . W (+ bias): one launch (pn_dense.hip); trans reads W^T from the same (C, K)-major... kernel
  int dense_plain(const float* x, int ldx, const float* Wk, int ldw, bool trans, int K, int C, const float* bias, float* out, int rows = -1) {
    return dense_layer(x, ldx, Wk, ldw, trans, rows < 0 ? Bd : rows, K, C, w.dense_part, w.dcount, bias, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0, 0,
                       nullptr, 1.f, out, nullptr, nullptr, nullptr, st);
  }
  int fwd_dense(DLs& dl, const LRef& r, const float* x, int act, const unsigned char* keep) {
    const int mode = r.has_bn ? (bn_batch(r.block) ? 1 : 2) : 0;
    const float ks = 1.f / (1.f - d.dropout_rate);
    return dense_layer(x, r.cin, p(r.kernel), r.cout, false, Bd, r.cin, r.cout, w.dense_part, w.dcount, p(r.bias), p(r.gamma), p(r.beta),
                       p(r.mm), p(r.mv), d.bn_momentum, d.bn_eps, mode, act, keep, ks, dl.z, dl.a, dl.mean, dl.invstd, st);
  }
  int fwd_tnet(TN& t, const TRef& r, const pn_operand* x) {
    if (fused_chain(t.c2) && (r.K == 3 || t.c1.wt16)) {
      PN_TRY(fwd_chain(r.K == 3 ? nullptr : x, r.K == 3 ? p(r.c1.kernel) : nullptr, t.c1, t.c2, t.c3, t.m3, r.c3));
      PN_TRY(fwd_dense(t.d1, r.d1, pooled(t.m3), 1, nullptr));
      PN_TRY(fwd_dense(t.d2, r.d2, t.d1.a, 1, nullptr));
      return dense_plain(t.d2.a, 256, p(r.w), r.K * r.K, false, 256, r.K * r.K, p(r.b), t.R);
    }
    if (r.K == 3) {
      PN_TRY(conv3_fwd(w.pcn, p(r.c1.kernel), 0, B, N, 64, t.c1.Z, bn_batch(r.c1.block) ? t.c1.part : nullptr, st, s16));
      PN_TRY(bn_fin(t.c1, r.c1));
    } else {
      PN_TRY(fwd_conv(t.c1, r.c1, *x, p(r.c1.kernel), 0, nullptr));
    }
    PN_TRY(fwd_conv(t.c2, r.c2, lazy(t.c1), p(r.c2.kernel), 0, nullptr));
    PN_TRY(fwd_max(t.c3, t.m3, r.c3, lazy(t.c2), r.K == 3 ? 0 : 1));
    PN_TRY(fwd_dense(t.d1, r.d1, pooled(t.m3), 1, nullptr));
    PN_TRY(fwd_dense(t.d2, r.d2, t.d1.a, 1, nullptr));
    return dense_plain(t.d2.a, 256, p(r.w), r.K * r.K, false, 256, r.K * r.K, p(r.b), t.R);      // (Bd, K, K): this rank's clouds at loc(t.R, K * K)
  }

  pn_operand x64op() const { return d.vanilla ? lazy(w.m12) : plain_act(w.X64, 64); }

  int forward() {
    {   // bf16 channel-major copies of the three 128->1024 kernels (they only change in the optimizer) + the dense layers' arrival counters
      const bool x3 = prec == PN_PREC_BF16X3;
      const float* ws[3] = {d.vanilla ? nullptr : p(L.iT.c3.kernel), d.vanilla ? nullptr : p(L.fT.c3.kernel), p(L.m23.kernel)};
      const int Ks[3] = {128, 128, 128}, Cs[3] = {1024, 1024, 1024};
      void* his[3] = {d.vanilla ? nullptr : w.iT.m3.wb_hi, d.vanilla ? nullptr : w.fT.m3.wb_hi, w.mm23.wb_hi};
      void* los[3] = {(x3 && !d.vanilla) ? w.iT.m3.wb_lo : nullptr, (x3 && !d.vanilla) ? w.fT.m3.wb_lo : nullptr, x3 ? w.mm23.wb_lo : nullptr};
      const float* sgs[3] = {d.vanilla ? nullptr : p(L.iT.c3.gamma), d.vanilla ? nullptr : p(L.fT.c3.gamma), p(L.m23.gamma)};
      // copies carry sign(gamma): max(sgn*z) needs no multiply.  The same launch normalises the clouds and, when the caller asks,
      // clears the gradient buffer and draws the dropout masks (pn_prologue.hip)
      // + the bf16 copies of the other per-point kernels for the row GEMMs (bf16 mode)
      WCopyDesc wc[PN_WCOPY_MAX];
      int nwc = 0;
      bool dropped = false;      // a layer that does not fit the launch's tables must fail the call, not run on stale copies / coefficients
      auto add_wc = [&](const CL& l, const float* Wk, int frag = 0) {
        if (!l.w16) return;
        if (nwc < PN_WCOPY_MAX) wc[nwc++] = WCopyDesc{Wk, l.w16, l.wt16, l.K, l.C, frag};
        else dropped = true;
      };
      if (!d.vanilla) {
        add_wc(w.iT.c2, p(L.iT.c2.kernel));
        add_wc(w.fT.c1, p(L.fT.c1.kernel));
        add_wc(w.fT.c2, p(L.fT.c2.kernel));
      }
      add_wc(w.m12, p(L.m12.kernel)); add_wc(w.m21, p(L.m21.kernel)); add_wc(w.m22, p(L.m22.kernel));
      // the fused frozen head reads its kernels as MFMA fragments (pn_segout.hip); the layer-by-layer plan as [C][K] rows
      const int sf = fused_seg_head() ? 1 : 0;
      add_wc(w.s1, p(L.s1.kernel), sf); add_wc(w.s2, p(L.s2.kernel), sf); add_wc(w.s3, p(L.s3.kernel), sf); add_wc(w.s4, p(L.s4.kernel), sf);
      // + the coefficients of every per-point layer whose BatchNormalization uses its moving statistics (bn_fin skips those layers)
      FrozenBnDesc fz[PN_FROZEN_MAX];
      int nfz = 0;
      auto add_fz = [&](const CL& l, const LRef& r) {
        if (bn_batch(r.block)) return;
        if (nfz < PN_FROZEN_MAX) fz[nfz++] = FrozenBnDesc{p(r.gamma), p(r.beta), p(r.mm), p(r.mv), l.mean, l.invstd, l.scale, l.shift, r.cout};
        else dropped = true;
      };
      if (!d.vanilla) {
        add_fz(w.iT.c1, L.iT.c1); add_fz(w.iT.c2, L.iT.c2); add_fz(w.fT.c1, L.fT.c1); add_fz(w.fT.c2, L.fT.c2);
      }
      add_fz(w.m11, L.m11); add_fz(w.m12, L.m12); add_fz(w.m21, L.m21); add_fz(w.m22, L.m22);
      add_fz(w.s1, L.s1); add_fz(w.s2, L.s2); add_fz(w.s3, L.s3); add_fz(w.s4, L.s4);
      if (dropped) {
        set_error("pn_model_forward: more per-point layers than fwd_prologue's tables hold (PN_WCOPY_MAX %d, PN_FROZEN_MAX %d)", PN_WCOPY_MAX,
                  PN_FROZEN_MAX);
        return PN_ERR_INVALID_ARGUMENT;
      }
      const bool zg = training && G && io.zero_grads_in_forward;
      const bool dm = training && io.dropout_step && io.keep1 && io.keep2 && d.dropout_rate > 0.f;
      PN_TRY(fwd_prologue(io.pc, B, N, w.pcn, w.cent, w.scl, ws, sgs, Ks, Cs, his, los, w.dcount, DENSE_MAX_COUNTERS + 3 * B * 512, zg ? G : nullptr,
                          zg ? L.total : 0, dm ? const_cast<unsigned char*>(io.keep1) : nullptr, dm ? (long long)Bd * 512 : 0,
                          dm ? const_cast<unsigned char*>(io.keep2) : nullptr, dm ? (long long)Bd * 256 : 0, d.dropout_rate, io.dropout_seed,
                          dm ? io.dropout_step : nullptr, wc, nwc, fz, nfz, d.bn_eps, st));
    }
    if (!d.vanilla) {
      PN_TRY(fwd_tnet(w.iT, L.iT, nullptr));
      // tf.matmul(pc, R) (PointNet.py:207) folded into mlp_1_1's kernel inside the launch; it also leaves W_eff and the third output
      PN_TRY(conv3_fwd(w.pcn, p(L.m11.kernel), 0, B, N, 64, w.m11.Z, bn_batch(BLK_M11) ? w.m11.part : nullptr, st, s16, loc(w.iT.R, 9), w.Weff1,
                       io.out_R));
    } else {
      PN_TRY(conv3_fwd(w.pcn, p(L.m11.kernel), 0, B, N, 64, w.m11.Z, bn_batch(BLK_M11) ? w.m11.part : nullptr, st, s16));
    }
    PN_TRY(bn_fin(w.m11, L.m11));
    PN_TRY(fwd_conv(w.m12, L.m12, lazy(w.m11), p(L.m12.kernel), 0, nullptr));
    if (!d.vanilla) {
      const pn_operand a12 = lazy(w.m12);
      PN_TRY(fwd_tnet(w.fT, L.fT, &a12));
      PN_TRY(conv_fwd(&a12, loc(w.fT.R, 4096), 4096, B, N, 64, 64, nullptr, w.X64, nullptr, prec, st));
    }
    const pn_operand x64 = x64op();
    if (fused_chain(w.m22) && w.m21.wt16) {
      PN_TRY(fwd_chain(&x64, nullptr, w.m21, w.m22, w.m23, w.mm23, L.m23));
    } else {
      PN_TRY(fwd_conv(w.m21, L.m21, x64, p(L.m21.kernel), 0, nullptr));
      PN_TRY(fwd_conv(w.m22, L.m22, lazy(w.m21), p(L.m22.kernel), 0, nullptr));
      PN_TRY(fwd_max(w.m23, w.mm23, L.m23, lazy(w.m22), 2));
    }
    const float* Gf = pooled(w.mm23);             // (Bd, 1024)

    // classification head (PointNet.py:252-263)
    {   // + in the same launch the global-feature half of seg_l1's kernel applied to the pooled vector (PointNet.py:268-275): w.gb
      const LRef& r = L.c1;
      const int mode = r.has_bn ? (bn_batch(r.block) ? 1 : 2) : 0;
      PN_TRY(dense_layer_with_plain(Gf, r.cin, p(r.kernel), r.cout, Bd, r.cin, r.cout, w.dense_part, w.dcount, p(r.bias), p(r.gamma), p(r.beta),
                                    p(r.mm), p(r.mv), d.bn_momentum, d.bn_eps, mode, 1, training ? io.keep1 : nullptr,
                                    1.f / (1.f - d.dropout_rate), w.c1.z, w.c1.a, w.c1.mean, w.c1.invstd, p(L.s1.kernel) + 64 * 512, 512, w.gb,
                                    st));
    }
    PN_TRY(fwd_dense(w.c2, L.c2, w.c1.a, 1, training ? io.keep2 : nullptr));
    PN_TRY(dense_plain(w.c2.a, 256, p(L.c3.kernel), d.ccls, false, 256, d.ccls, p(L.c3.bias), w.cls_logits));
    const bool fused = io.labels_cls != nullptr;     // the softmax + loss of these logits: in the pass's last launch, below

    // segmentation head (PointNet.py:268-290)
    const bool fseg = io.labels_seg != nullptr;
    int seg_parts = (int)cdivll(M, seg_out_part_rows());
    if (fused_seg_head()) {
      // frozen head, no gradient through it: one launch, activations stay on chip (pn_segout.hip: seg_head_fused)
      seg_parts = B * cdiv(N, seg_head_fused_rows());
      PN_TRY(seg_head_fused(&x64, loc(w.gb, 512), w.s1.wt16, w.s2.wt16, w.s3.wt16, w.s4.wt16, w.s1.scale, w.s1.shift, w.s2.scale, w.s2.shift, w.s3.scale,
                            w.s3.shift, w.s4.scale, w.s4.shift, p(L.s5.kernel), p(L.s5.bias), B, N, d.cseg, s16, io.labels_seg,
                            fseg ? io.loss_weights[1] / (float)M : 0.f, io.out_seg, nullptr, fseg ? w.seg_part : nullptr, st));
    } else {
    const float* Ws1 = p(L.s1.kernel);
    // seg_l1 always emits its forward partials: the backward needs the per-cloud sums of z
    PN_TRY(conv_fwd(&x64, Ws1, 0, B, N, 64, 512, loc(w.gb, 512), w.s1.Z, w.s1.part, prec, st, w.s1.wt16));
    PN_TRY(bn_fin(w.s1, L.s1));
    PN_TRY(fwd_conv(w.s2, L.s2, lazy(w.s1), p(L.s2.kernel), 0, nullptr));
    PN_TRY(fwd_conv(w.s3, L.s3, lazy(w.s2), p(L.s3.kernel), 0, nullptr));
    PN_TRY(fwd_conv(w.s4, L.s4, lazy(w.s3), p(L.s4.kernel), 0, nullptr));
    const pn_operand a4 = lazy(w.s4);
    PN_TRY(seg_out_fwd(&a4, p(L.s5.kernel), p(L.s5.bias), M, 128, d.cseg, io.labels_seg, fseg ? io.loss_weights[1] / (float)M : 0.f,
                       io.out_seg, (fseg && training) ? w.seg_dlogits : nullptr, fseg ? w.seg_part : nullptr, st));
    }
    // third output: the input transform (PointNet.py:292); identity for vanilla (:211)
    if (io.out_R) {
      if (d.vanilla) {
        PN_TRY(fill_eye3(io.out_R, B, st));
      }
    }
    {   // one launch: classification softmax (+ loss, d logits), the segmentation loss / accuracy sums, the rotation loss value
      const float* Rp = (io.se3 && io.scalars) ? (d.vanilla ? io.out_R : loc(w.iT.R, 9)) : nullptr;
      const bool sums = fseg && io.scalars;
      PN_TRY(loss_tail(loc(w.cls_logits, d.ccls), B, d.ccls, io.labels_cls, fused ? io.loss_weights[0] / (float)B : 0.f, io.out_cls,
                       (fused && training) ? loc(w.cls_dlogits, d.ccls) : nullptr, io.scalars ? io.scalars + 0 : nullptr,
                       io.scalars ? io.scalars + 1 : nullptr, sums ? w.seg_part : nullptr, sums ? seg_parts : 0,
                       seg_out_part_stride(), sums ? 2 : 0, sums ? io.scalars + 2 : nullptr, Rp, io.se3, B * 9, Rp ? io.scalars + 4 : nullptr,
                       st));
    }
    if (io.scalars && !d.vanilla) {
      if (d.reg_in) {
        PN_TRY(orth_reg(loc(w.iT.R, 9), B, 3, 1e-3f, nullptr, w.regpart, st));
        PN_TRY(sum_partials(w.regpart, B, 1, 1, io.scalars + 5, st));
      }
      if (d.reg_feat) {
        PN_TRY(orth_reg(loc(w.fT.R, 4096), B, 64, 1e-3f, nullptr, w.regpart + B, st));
        PN_TRY(sum_partials(w.regpart + B, B, 1, 1, io.scalars + 6, st));
      }
    }
    return PN_OK;
  }

  // ---------------- backward pieces ----------------
  int wgrad_to(const pn_operand& a, const pn_operand& b, int Ci, int Cj, float* out, bool per_cloud, bool deferrable = false) {
    return wgrad_general(a, b, B, N, Ci, Cj, out, per_cloud, prec, false, deferrable);
  }
  // parameter gradient of a Cin = 3 layer from T row-tile slabs
  int wgrad3_to(const pn_operand& dz, float* out) {
    float* sl = pool_take((size_t)T * 3 * 64);
    if (sl) {
      PN_TRY(conv3_wgrad(w.pcn, &dz, B, N, 64, sl, st));
      jobs.push_back(SlabJob{sl, out, 3 * 64, T});
      return PN_OK;
    }
    PN_TRY(conv3_wgrad(w.pcn, &dz, B, N, 64, cur_slabs(), st));
    return slab_reduce(cur_slabs(), T, T, 3 * 64, out, st);
  }
  // colsum: `out` receives Ci*Cj products followed by Ci column sums of operand a (Gram matrix + a1 in one pass over the rows)
  int wgrad_general(const pn_operand& a, const pn_operand& b, int Bq, int Nq, int Ci, int Cj, float* out, bool per_cloud, int pr,
                    bool colsum = false, bool deferrable = false) {
    // deferred jobs share their launch with others of the same tile shape, so each can do with fewer, longer slabs (never more than
    // the plan sized the pool for: the default target is the upper bound)
    // the three Gram matrices in one launch: 128 slabs each (1.063 -> 1.053 ms/step); the 64-wide jobs measured the same at 128 and 256
    static const int t_gram = getenv("PN_WGRAD_TARGET_GRAM") ? atoi(getenv("PN_WGRAD_TARGET_GRAM")) : 128;
    const int t_over = (deferrable && colsum) ? t_gram : 0;
    int spc;
    const int rows = (int)wgrad_slab_rows(Bq, Nq, Ci, Cj, &spc, t_over);
    const size_t elems = (size_t)Ci * Cj + (colsum ? Ci : 0);
    last_deferred = false;
    if (deferrable && !per_cloud) {
      if (float* ps = pool_take((size_t)Bq * spc * elems)) {
        static const bool batch_gemm = !(getenv("PN_WGRAD_BATCH") && atoi(getenv("PN_WGRAD_BATCH")) == 0);
        if (batch_gemm) wg_jobs.push_back(WgradDesc{a, b, Bq, Nq, Ci, Cj, rows, ps, pr, colsum ? 1 : 0, 0});
        else PN_TRY(conv_wgrad(&a, &b, Bq, Nq, Ci, Cj, rows, ps, pr, st, colsum ? 1 : 0));
        jobs.push_back(SlabJob{ps, out, (long long)elems, Bq * spc});
        last_deferred = true;
        return PN_OK;
      }
    }
    float* sl = cur_slabs();
    if ((size_t)Bq * spc * elems > (sl == w.slabs ? w.slab_floats : w.slab_main_floats)) {
      set_error("wgrad: slab scratch too small");
      return PN_ERR_WORKSPACE;
    }
    PN_TRY(conv_wgrad(&a, &b, Bq, Nq, Ci, Cj, rows, sl, pr, st, colsum ? 1 : 0));
    return slab_reduce(sl, Bq * spc, per_cloud ? spc : Bq * spc, (long long)elems, out, st);
  }
  int bn_bwd_fin(const CL& l, const LRef& r, const float* part) {
    const int bs = bn_batch(r.block) ? 1 : 0;
    if (W > 1 && bs) {           // sum dy_hat, sum dy_hat * z over every rank's rows (out of place: seg_l1's own partials feed cloud_bias_grad)
      PN_TRY(sync_sum(part, w.sync_part, (long long)T * 2 * r.cout));
      part = w.sync_part;
    }
    return bn_bwd_finalize(part, T, r.cout, M * W, p(r.gamma), l.mean, l.invstd, bs, bs ? gr(r.gamma) : nullptr, bs ? gr(r.beta) : nullptr,
                           l.ca, l.cb, l.cc, st);
  }
  // standard interior step: given cur.dy (+ w.bpart holding its stats) produce prev.dy and cur's weight gradient
  int bwd_step(CL& cur, const LRef& rc, CL& prev, const pn_operand& prev_act) {
    PN_TRY(bn_bwd_fin(cur, rc, w.bpart));
    const pn_operand dz = dzop(cur);
    if (tr(rc.block) && G) {
      float* out = gr(rc.kernel);
      const int ci = rc.cin, cj = rc.cout;
      PN_TRY(side([=] { return wgrad_to(prev_act, dz, ci, cj, out, false, true); }));
      PN_TRY(flush());
    }
    return conv_bwd_data(&dz, p(rc.kernel), 0, B, N, rc.cout, rc.cin, nullptr, prev.Z, prev.scale, prev.shift, prev.dy, w.bpart, prec, st,
                         cur.w16);
  }
  // backward of a max-pooled layer: dG (B,C) -> prev.dy (+stats in w.bpart), this layer's parameter gradients
  int bwd_max(CL& l, ML& m, const LRef& r, const pn_operand& xop, CL& prev, const float* dG, const float* dG2 = nullptr) {
    const int K = r.cin, C = r.cout;
    const int bs = bn_batch(r.block) ? 1 : 0;
    const bool wg = tr(r.block) && G;
    // Pm in the preparation launch itself (PN_PM_IN_PREP=0: the weight-gradient launch of rounds 1-2)
    static const bool pm_in_prep = !(getenv("PN_PM_IN_PREP") && atoi(getenv("PN_PM_IN_PREP")) == 0);
    float* pms = (pm_in_prep && K == 128 && C % 32 == 0 && C / 32 <= 32) ? w.pm_slabs : nullptr;
    // + the channel-major copies Wt, We = -e (.) Wt used below, and -- in the same launch, on workgroups of their own -- the rows of
    // the maxima (m.argq, left by the forward pass, -> m.arg: pn_maxbwd.hip)
    if (W > 1) {
      // synchronised BN: dG / dG2 hold every rank's clouds (Bd rows), and so do the pooled maxima: hs for all of them, the batch terms
      // e, f (and dgamma, dbeta) from the sums over ALL clouds; the rows of the maxima are resolved for this rank's clouds only
      PN_TRY(maxbwd_prep(dG, dG2, m.g_all, m.zstar_all, Bd, C, l.mean, l.invstd, l.scale, bs, M * W, m.hs, m.e, m.nege, m.f, wg ? gr(r.gamma) : nullptr,
                         wg ? gr(r.beta) : nullptr, p(r.kernel), K, m.Wt, m.We, st, pms));
      PN_TRY(max_resolve(&xop, m.wb_hi, m.wb_lo, prec, m.argq, B, N, K, C, m.arg, st));
    } else {
    PN_TRY(maxbwd_prep_resolve(dG, dG2, m.g, m.zstar, B, C, l.mean, l.invstd, l.scale, bs, M, m.hs, m.e, m.nege, m.f, wg ? gr(r.gamma) : nullptr,
                               wg ? gr(r.beta) : nullptr, p(r.kernel), K, m.Wt, m.We, &xop, m.wb_hi, m.wb_lo, prec, m.argq, N, m.arg, st, pms));
    }
    const float* hs_loc = W > 1 ? loc(m.hs, C) : m.hs;       // this rank's clouds
    // the parameter-gradient branch forks here: it needs m.arg, hs, e, f of the launch above and nothing of what follows
    if (wg) {
      ML mm = m;
      mm.hs = const_cast<float*>(hs_loc);
      float* dw = gr(r.kernel);
      const float* Wk = p(r.kernel);
      // this whole branch feeds only dW: the Gram slabs are reduced with the other deferred jobs and the two consumers follow them
      PN_TRY(side([=] {
        if (gram_rides(mm, r)) {       // the slabs were written in the forward pass (fwd_max): only their reduction is left, with the others
          jobs.push_back(SlabJob{mm.gram_slabs, mm.gram, (long long)K * K + K, B * mm.gram_spc});
          last_deferred = true;
        } else
        PN_TRY(wgrad_general(xop, xop, B, N, K, K, mm.gram, false, prec, true, true));   // Gram matrix and a1 = column sums together
        // G W in weight-gradient form: out[k][c] = sum_k' G[k'][k] W[k'][c] over ONE slab of K rows written straight into GW (G is
        // symmetric up to the rounding of its cross terms); in that form the three layers can share a launch (conv_wgrad_batch)
        const WgradDesc gwd{plain(mm.gram, K), plain(Wk, C), 1, K, K, C, K, mm.GW, PN_PREC_BF16X3, 0, 1};   // 64x64 tiles: 32 workgroups per layer
        auto gw_alone = [=] { return conv_wgrad_batch(&gwd, 1, st); };
        auto rest = [=] {
          PN_TRY(gw_alone());
          return maxbwd_dw(&xop, mm.arg, mm.hs, B, N, K, C, mm.a1, mm.f, mm.e, mm.GW, dw, st);
        };
        if (!last_deferred) return rest();
        if (!dw_jobs.empty() && (dw_K != K || dw_C != C)) {     // another shape than the batch so far: keep this layer's own launch
          after_jobs.push_back(rest);
          return (int)PN_OK;
        }
        static const bool gw_batch = !(getenv("PN_GW_BATCH") && atoi(getenv("PN_GW_BATCH")) == 0);
        if (gw_batch) gw_jobs.push_back(gwd);
        else after_jobs.push_back(gw_alone);
        dw_K = K; dw_C = C;
        dw_jobs.push_back(DwJob{xop, mm.arg, mm.hs, mm.a1, mm.f, mm.e, mm.GW, dw});
        return (int)PN_OK;
      }));
      PN_TRY(flush());
    }
    // Pm[k'][k] = sum_c (-e_c) W[k'][c] W[k][c]: the contraction runs over the 1024 channels, so it is laid out as a
    // weight-gradient problem over "rows" c (16 slabs of 64 channels -> 16 workgroups) instead of one 128x128 tile
    // ... and q = W f (needs only maxbwd_prep's f) rides in the launch that reduces those slabs.
    // Round 3: the preparation workgroups form their 32 channels' share of Pm themselves (pms: C / 32 slabs) and the launch is gone.
    if (pms) {
      // ... and the reduction + q ride behind the scatter's workgroups; q enters the data gradient as the GEMM's per-column constant
      PN_TRY(maxbwd_scatter_reduce(m.arg, hs_loc, m.Wt, B, N, K, C, m.D, s16, pms, C / 32, (long long)K * K, m.Pm, p(r.kernel), m.f, m.q, st));
      return conv_bwd_data(&xop, m.Pm, 0, B, N, K, K, m.D, prev.Z, prev.scale, prev.shift, prev.dy, w.bpart, prec, st, nullptr, m.q);
    } else {
      int spc;
      const int rows = (int)wgrad_slab_rows(1, C, K, K, &spc);
      float* sl = cur_slabs();
      if ((size_t)spc * K * K > (sl == w.slabs ? w.slab_floats : w.slab_main_floats)) {
        set_error("wgrad: slab scratch too small");
        return PN_ERR_WORKSPACE;
      }
      const pn_operand we = plain(m.We, K), wt = plain(m.Wt, K);
      static const bool pm_small = !(getenv("PN_PM_SMALL") && atoi(getenv("PN_PM_SMALL")) == 0);
      const WgradDesc pmd{we, wt, 1, C, K, K, rows, sl, PN_PREC_BF16X3, 0, pm_small ? 1 : 0};   // 64x64 tiles: 4x the workgroups of this 16-slab job
      PN_TRY(conv_wgrad_batch(&pmd, 1, st));
      PN_TRY(slab_reduce_q(sl, spc, (long long)K * K, m.Pm, p(r.kernel), m.f, K, C, m.q, st));
    }
    PN_TRY(maxbwd_scatter(m.arg, hs_loc, m.Wt, m.q, B, N, K, C, m.D, s16, st));
    PN_TRY(conv_bwd_data(&xop, m.Pm, 0, B, N, K, K, m.D, prev.Z, prev.scale, prev.shift, prev.dy, w.bpart, prec, st));
    return PN_OK;
  }
  // dense layer backward: da (B,C) -> dx (B,K) written to dx_out; parameter gradients
  int bwd_dense(DLs& dl, const LRef& r, const float* xin, const float* da, int act, const unsigned char* keep, float* dx_out) {
    const int mode = r.has_bn ? (bn_batch(r.block) ? 1 : 2) : 0;
    const bool wg = tr(r.block) && G;
    const float ks = 1.f / (1.f - d.dropout_rate);
    float* dgam = (wg && mode == 1) ? gr(r.gamma) : nullptr;
    float* dbet = (wg && mode == 1) ? gr(r.beta) : nullptr;
    float* dbia = (wg && mode == 0) ? gr(r.bias) : nullptr;
    if (Bd <= 32) {
      // dropout/relu/BN backward fused into the weight gradient: one launch
      PN_TRY(dense_bwd_fused(da, dl.z, xin, r.cin, Bd, r.cin, r.cout, p(r.gamma), p(r.beta), dl.mean, dl.invstd, mode, act, keep, ks, dl.dz,
                             dgam, dbet, dbia, wg ? gr(r.kernel) : nullptr, st));
    } else {
      PN_TRY(dense_bwd_pre(da, dl.z, Bd, r.cout, p(r.gamma), p(r.beta), dl.mean, dl.invstd, mode, act, keep, ks, dl.dz, dgam, dbet, dbia, st));
      if (wg) {
        const float* dzp = dl.dz;
        float* out = gr(r.kernel);
        const int ci = r.cin, cj = r.cout;
        const int rows = Bd;
        PN_TRY(side([=] { return dense_wgrad(xin, ci, dzp, rows, ci, cj, out, st); }));
        PN_TRY(flush());
      }
    }
    // dx = dz . W^T, read from the layer's own (K, C) kernel
    if (dx_out) PN_TRY(dense_plain(dl.dz, r.cout, p(r.kernel), r.cout, true, r.cout, r.cin, nullptr, dx_out));
    return PN_OK;
  }
  // ---- a chain of per-cloud dense layers taken backward with ONE launch per layer (rows = B <= 32, no auxiliary stream) ----------
  // dtop (B, Ctop) is d(output) of the chain's top product, out = a_below . Wtop (+ bias), which has no BatchNormalization / ReLU of
  // its own (the logits layer, the T-Net's K x K output).  Every launch is a TRANS product dx = dz . W^T whose finishing workgroups go
  // straight on through the layer below (dropout -> ReLU -> BatchNormalization backward are per column): dz of that layer, dgamma,
  // dbeta.  The weight gradients x^T . dz (and the top bias gradient) are nobody's input before the optimizer: collected in
  // dense_jobs, one launch per pass (flush_jobs).  Round 2's form took two launches per layer (dz + dW, then dx).
  struct ChainLayer { DLs* dl; const LRef* r; const float* xin; int act; const unsigned char* keep; };
  bool chain_ok() const { return Bd <= 32; }      // (with or without an auxiliary stream: both step layouts must give the same bits)
  int bwd_chain(const float* dtop, int Ctop, const float* Wtop, const float* a_below_top, float* dWtop, float* dbtop, float* da_top,
                ChainLayer* ls, int n, float* dx_out) {
    // top product: its own weight gradient is a plain job on dtop
    if (dWtop) dense_jobs.push_back(DenseWgradJob{a_below_top, ls[0].r->cout, dtop, Bd, ls[0].r->cout, Ctop, dWtop, dbtop});
    const float* dz_above = dtop;
    int c_above = Ctop;
    const float* w_above = Wtop;
    float* dx_above = da_top;                        // where d(activation) of layer 0 goes (kept: debugging, tests)
    const float ks = 1.f / (1.f - d.dropout_rate);
    for (int q = 0; q < n; ++q) {
      DLs& dl = *ls[q].dl;
      const LRef& r = *ls[q].r;
      const int mode = r.has_bn ? (bn_batch(r.block) ? 1 : 2) : 0;
      const bool wg = tr(r.block) && G;
      DenseTail t;
      memset(&t, 0, sizeof(t));
      t.z = dl.z; t.gamma = p(r.gamma); t.beta = p(r.beta); t.mean = dl.mean; t.invstd = dl.invstd;
      t.keep = ls[q].keep; t.keep_scale = ks; t.mode = mode; t.act = ls[q].act;
      t.dz = dl.dz;
      t.dgamma = (wg && mode == 1) ? gr(r.gamma) : nullptr;
      t.dbeta = (wg && mode == 1) ? gr(r.beta) : nullptr;
      t.dbias = (wg && mode == 0) ? gr(r.bias) : nullptr;
      PN_TRY(dense_trans_tail(dz_above, c_above, w_above, c_above, Bd, c_above, r.cout, w.dense_part, w.dcount, dx_above, &t, st));
      if (wg) dense_jobs.push_back(DenseWgradJob{ls[q].xin, r.cin, dl.dz, Bd, r.cin, r.cout, gr(r.kernel), nullptr});
      dz_above = dl.dz; c_above = r.cout; w_above = p(r.kernel); dx_above = dl.din;
    }
    // below the last layer: a plain product
    return dense_trans_tail(dz_above, c_above, w_above, c_above, Bd, c_above, ls[n - 1].r->cin, w.dense_part, w.dcount, dx_out, nullptr, st);
  }
  // T-Net backward from dR (B,K*K); leaves c1's dz coefficients ready (c1.dy + c1.ca/cb/cc)
  int bwd_tnet(TN& t, const TRef& r, const pn_operand* x) {
    const int KK = r.K * r.K;
    const bool wg = tr(r.c1.block) && G;
    PN_TRY(sync_gather_rows(t.dR, (long long)B * KK));      // synchronised BN: the tail runs backward on every rank's clouds
    if (chain_ok()) {
      ChainLayer ls[2] = {{&t.d2, &r.d2, t.d1.a, 1, nullptr}, {&t.d1, &r.d1, pooled(t.m3), 1, nullptr}};
      PN_TRY(bwd_chain(t.dR, KK, p(r.w), t.d2.a, wg ? gr(r.w) : nullptr, wg ? gr(r.b) : nullptr, t.da2, ls, 2, t.m3.dG));
    } else {
    if (wg) {
      const float *dR = t.dR, *a2 = t.d2.a;
      float *gb = gr(r.b), *gw = gr(r.w);
      const int rows = Bd;
      PN_TRY(side([=] {
        return dense_wgrad(a2, 256, dR, rows, 256, KK, gw, st, gb);        // dw = a2^T dR and db = sum_b dR in one launch
      }));
      PN_TRY(flush());
    }
    PN_TRY(dense_plain(t.dR, KK, p(r.w), KK, true, KK, 256, nullptr, t.da2));
    PN_TRY(bwd_dense(t.d2, r.d2, t.d1.a, t.da2, 1, nullptr, t.d2.din));
    PN_TRY(bwd_dense(t.d1, r.d1, pooled(t.m3), t.d2.din, 1, nullptr, t.m3.dG));
    }
    PN_TRY(bwd_max(t.c3, t.m3, r.c3, lazy(t.c2), t.c2, t.m3.dG));
    // c2 -> c1
    PN_TRY(bwd_step(t.c2, r.c2, t.c1, lazy(t.c1)));
    PN_TRY(bn_bwd_fin(t.c1, r.c1, w.bpart));
    const pn_operand dz1 = dzop(t.c1);
    if (wg) {
      float* out = gr(r.c1.kernel);
      if (r.K == 3) {
        PN_TRY(side([=] { return wgrad3_to(dz1, out); }));
      } else {
        const pn_operand xin = *x;
        PN_TRY(side([=] { return wgrad_to(xin, dz1, 64, 64, out, false, true); }));
      }
      PN_TRY(flush());
    }
    return PN_OK;
  }

  int backward(const float* d_cls, const float* d_seg, const float* d_R) {
    const int rc = backward_body(d_cls, d_seg, d_R);
    if (rc != PN_OK) return rc;
    return flush_jobs();                       // every deferred parameter gradient of this pass (or phase of it) is final after this
  }
  int backward_body(const float* d_cls, const float* d_seg, const float* d_R) {
    if (!G) {
      set_error("pn_model_backward: grads buffer is NULL");
      return PN_ERR_INVALID_ARGUMENT;
    }
    const pn_operand x64 = x64op();
    const float* Ws1 = p(L.s1.kernel);
    const bool fused = io.labels_cls != nullptr || io.labels_seg != nullptr;
    bool has_seg = d_seg != nullptr || (io.labels_seg != nullptr && io.loss_weights[1] != 0.f);
    if (has_seg && fused_seg_head()) {
      set_error("pn_model_backward: a gradient through the segmentation head was asked for, but the forward pass ran the head fused "
                "(frozen, loss weight 0) and kept none of its activations; set PN_IO_KEEP_ACTIVATIONS in pn_model_io.flags");
      return PN_ERR_INVALID_ARGUMENT;
    }
    bool has_cls = d_cls != nullptr || (io.labels_cls != nullptr && io.loss_weights[0] != 0.f);
    (void)fused;
    const bool have_R_grad = !d.vanilla && (d_R != nullptr || (io.se3 != nullptr && io.loss_weights[2] != 0.f) || d.reg_in);
    const bool trunk = has_seg || has_cls || (!d.vanilla && d.reg_feat);
    // pn_model_io.bwd_phase: 0 = the whole backward pass; 1 = heads, mlp_2 and the feature transform (every gradient slot from
    // feature_transform.* to the end of the flat buffer is final afterwards); 2 = mlp_1 and the input transform (the slots before
    // feature_transform.*).  A data-parallel caller all-reduces the first bucket while phase 2 runs (engine.TrainStep).
    const int phase = io.bwd_phase;
    if (phase != 2) {
    if (!io.zero_grads_in_forward) PN_TRY(zero_fill2(G, L.total, reinterpret_cast<float*>(w.dcount), DENSE_MAX_COUNTERS, st));

    // ---- segmentation head ----
    bool have_dx64 = false;     // w.dX64 holds the seg head's contribution to d(X_64)
    bool have_dGseg = false;
    if (has_seg) {
      if (d_seg) PN_TRY(softmax_bwd_rows(io.out_seg, d_seg, M, d.cseg, w.seg_dlogits, st));
      const pn_operand a4 = lazy(w.s4);
      PN_TRY(seg_out_bwd(&a4, p(L.s5.kernel), w.seg_dlogits, B, N, 128, d.cseg, w.s4.dy, w.bpart, w.s5slab, st, s16));
      if (tr(BLK_S5)) {
        PN_TRY(side([=] {
          if (!aux) jobs.push_back(SlabJob{w.s5slab, gr(L.s5.kernel), (long long)128 * d.cseg, T});     // s5slab is this job's own region
          else PN_TRY(slab_reduce(w.s5slab, T, T, (long long)128 * d.cseg, gr(L.s5.kernel), st));
          // bias gradient = column sums of dlogits: the forward's per-block partials already hold them when it produced dlogits itself
          if (!d_seg) return sum_partials(w.seg_part + 2, (int)cdivll(M, seg_out_part_rows()), seg_out_part_stride(), d.cseg, gr(L.s5.bias), st);
          return sum_partials(w.seg_dlogits, (int)M, d.cseg, d.cseg, gr(L.s5.bias), st);
        }));
        PN_TRY(flush());
      }
      PN_TRY(bwd_step(w.s4, L.s4, w.s3, lazy(w.s3)));
      PN_TRY(bwd_step(w.s3, L.s3, w.s2, lazy(w.s2)));
      PN_TRY(bwd_step(w.s2, L.s2, w.s1, lazy(w.s1)));
      PN_TRY(bn_bwd_fin(w.s1, L.s1, w.bpart));
      const pn_operand dz1 = dzop(w.s1);
      PN_TRY(cloud_bias_grad(w.bpart, w.s1.part, B, tpc, N, 512, w.s1.ca, w.s1.cb, w.s1.cc, w.dgb, st));
      if (tr(BLK_S1)) {
        PN_TRY(side([=] {
          PN_TRY(wgrad_to(x64, dz1, 64, 512, gr(L.s1.kernel), false, true));
          return dense_wgrad(pooled_local(w.mm23, 1024), 1024, w.dgb, B, 1024, 512, gr(L.s1.kernel) + 64 * 512, st);
        }));
        PN_TRY(flush());
      }
      PN_TRY(dense_plain(w.dgb, 512, Ws1 + 64 * 512, 512, true, 512, 1024, nullptr, loc(w.dGseg, 1024), B));      // this rank's clouds
      PN_TRY(sync_gather_rows(w.dGseg, (long long)B * 1024));
      have_dGseg = true;
      PN_TRY(conv_bwd_data(&dz1, Ws1, 0, B, N, 512, 64, nullptr, nullptr, nullptr, nullptr, w.dX64, nullptr, prec, st, w.s1.w16));
      have_dx64 = true;
    }

    // ---- classification head ----
    bool have_dGcls = false;
    if (has_cls) {
      if (d_cls) PN_TRY(softmax_bwd_rows(io.out_cls, d_cls, B, d.ccls, loc(w.cls_dlogits, d.ccls), st));
      PN_TRY(sync_gather_rows(w.cls_dlogits, (long long)B * d.ccls));      // synchronised BN: the head runs backward on every rank's clouds
      if (chain_ok()) {
        // the logits layer is the chain's top product (bias, no BatchNormalization, no activation: dz = d logits)
        const bool wg3 = tr(BLK_C3) && G;
        ChainLayer ls[2] = {{&w.c2, &L.c2, w.c1.a, 1, io.keep2}, {&w.c1, &L.c1, pooled(w.mm23), 1, io.keep1}};
        PN_TRY(bwd_chain(w.cls_dlogits, d.ccls, p(L.c3.kernel), w.c2.a, wg3 ? gr(L.c3.kernel) : nullptr, wg3 ? gr(L.c3.bias) : nullptr,
                         w.c3.din, ls, 2, w.dGcls));
      } else {
      PN_TRY(bwd_dense(w.c3, L.c3, w.c2.a, w.cls_dlogits, 0, nullptr, w.c3.din));
      PN_TRY(bwd_dense(w.c2, L.c2, w.c1.a, w.c3.din, 1, io.keep2, w.c2.din));
      PN_TRY(bwd_dense(w.c1, L.c1, pooled(w.mm23), w.c2.din, 1, io.keep1, w.dGcls));
      }
      have_dGcls = true;
    }
    // d(global feature) = classification-head part + segmentation-head part: summed inside maxbwd_prep
    const float* dGa = have_dGcls ? w.dGcls : nullptr;
    const float* dGb = have_dGseg ? w.dGseg : nullptr;

    if (!trunk && !have_R_grad) return PN_OK;

    // ---- mlp_2 ----
    if (has_seg || has_cls) {
      PN_TRY(bwd_max(w.m23, w.mm23, L.m23, lazy(w.m22), w.m22, dGa, dGb));
      PN_TRY(bwd_step(w.m22, L.m22, w.m21, lazy(w.m21)));
      PN_TRY(bn_bwd_fin(w.m21, L.m21, w.bpart));
      const pn_operand dz21 = dzop(w.m21);
      if (tr(BLK_M21)) {
        PN_TRY(side([=] { return wgrad_to(x64, dz21, 64, 64, gr(L.m21.kernel), false, true); }));
        PN_TRY(flush());
      }
      if (d.vanilla) {
        PN_TRY(conv_bwd_data(&dz21, p(L.m21.kernel), 0, B, N, 64, 64, have_dx64 ? w.dX64 : nullptr, w.m12.Z, w.m12.scale, w.m12.shift,
                             w.m12.dy, w.bpart, prec, st, w.m21.w16));
      } else {
        PN_TRY(conv_bwd_data(&dz21, p(L.m21.kernel), 0, B, N, 64, 64, have_dx64 ? w.dX64 : nullptr, nullptr, nullptr, nullptr, w.dX64,
                             nullptr, prec, st, w.m21.w16));
      }
    }
    if (!d.vanilla) {
      // ---- feature transform: X_64 = A_12 . R_64 ----
      const pn_operand a12 = lazy(w.m12);
      const bool have_dx = has_seg || has_cls;
      float* fdR = loc(w.fT.dR, 4096);                     // this rank's clouds (bwd_tnet gathers the others')
      const float* fR = loc(w.fT.R, 4096);
      if (!have_dx) PN_TRY(zero_fill(fdR, (long long)B * 4096, st));     // otherwise the slab reduction below is its first writer
      if (have_dx) {
        const pn_operand dx = plain_act(w.dX64, 64);
        PN_TRY(wgrad_to(a12, dx, 64, 64, fdR, true));
        PN_TRY(conv_bwd_data(&dx, fR, 4096, B, N, 64, 64, nullptr, nullptr, nullptr, nullptr, w.tmpA12, nullptr, prec, st));
      }
      if (d.reg_feat) PN_TRY(orth_reg(fR, B, 64, 1e-3f, fdR, nullptr, st));
      PN_TRY(bwd_tnet(w.fT, L.fT, &a12));
      const pn_operand dzf1 = dzop(w.fT.c1);
      PN_TRY(conv_bwd_data(&dzf1, p(L.fT.c1.kernel), 0, B, N, 64, 64, have_dx ? w.tmpA12 : nullptr, w.m12.Z, w.m12.scale, w.m12.shift,
                           w.m12.dy, w.bpart, prec, st, w.fT.c1.w16));
    }
    }   // phase != 2
    if (phase == 1) return PN_OK;
    if (!trunk && !have_R_grad) return PN_OK;
    // ---- mlp_1 ----
    PN_TRY(bwd_step(w.m12, L.m12, w.m11, lazy(w.m11)));
    PN_TRY(bn_bwd_fin(w.m11, L.m11, w.bpart));
    const pn_operand dz11 = dzop(w.m11);
    if (d.vanilla) {
      if (tr(BLK_M11)) PN_TRY(wgrad3_to(dz11, gr(L.m11.kernel)));
      return PN_OK;
    }
    PN_TRY(conv3_wgrad(w.pcn, &dz11, B, N, 64, cur_slabs(), st));
    // d(W_eff) per cloud = sum of its tiles' slabs; dR and dW of W_eff[b] = R[b] W follow in the same launch
    float* idR = loc(w.iT.dR, 9);
    const float* iR = loc(w.iT.R, 9);
    PN_TRY(fold3_bwd_slabs(cur_slabs(), T, tpc, iR, p(L.m11.kernel), B, 64, idR, tr(BLK_M11) ? gr(L.m11.kernel) : nullptr, st));
    // ---- input transform ----
    if (d_R) PN_TRY(axpy(d_R, 1.f, idR, (long long)B * 9, st));
    if (io.se3 && io.loss_weights[2] != 0.f && !d_R)
      PN_TRY(mse(iR, io.se3, B * 9, 2.f * io.loss_weights[2] / (float)(B * 9), idR, nullptr, st));
    if (d.reg_in) PN_TRY(orth_reg(iR, B, 3, 1e-3f, idR, nullptr, st));
    return bwd_tnet(w.iT, L.iT, nullptr);
  }
};

static int check_desc(const pn_model_desc* d) {
  PN_CHECK_ARG(d != nullptr, "pn_model: null descriptor");
  PN_CHECK_ARG(d->ccls >= 1 && d->ccls <= 4096, "pn_model: classification width %d out of range", d->ccls);
  PN_CHECK_ARG(d->cseg >= 1 && d->cseg <= 16, "pn_model: segmentation width %d not in [1,16]", d->cseg);
  PN_CHECK_ARG(d->dropout_rate >= 0.f && d->dropout_rate < 1.f, "pn_model: dropout rate must be in [0,1)");
  PN_CHECK_ARG(d->sync_world >= 0 && d->sync_world <= 64, "pn_model: sync_world %d out of range", d->sync_world);
  PN_CHECK_ARG(d->prec == PN_PREC_BF16 || d->prec == PN_PREC_BF16X3 || d->prec == (PN_PREC_BF16 | PN_STORE_BF16),
               "pn_model: bad prec %d (PN_PREC_BF16, PN_PREC_BF16X3 or PN_PREC_BF16 | PN_STORE_BF16)", d->prec);
  return PN_OK;
}

static int make_run(const pn_model_desc* d, const pn_model_io* io, hipStream_t st, Run*& out) {
  PN_TRY(check_desc(d));
  PN_CHECK_ARG(io && io->pc && io->params && io->workspace, "pn_model: null io pointer");
  PN_CHECK_ARG(io->B > 0 && io->N > 0, "pn_model: B and N must be positive (B=%d N=%d)", io->B, io->N);
  PN_CHECK_ARG(io->out_cls && io->out_seg, "pn_model: output buffers are required");
  Run* r = new Run{*d, *io};
  r->L = make_layout(*d);
  r->B = io->B; r->N = io->N; r->M = (long long)io->B * io->N;
  r->tpc = cdiv(io->N, 128); r->T = io->B * r->tpc;
  r->W = (io->training && d->sync_world > 1) ? d->sync_world : 1;
  r->rk = r->W > 1 ? io->sync_rank : 0;
  r->Bd = io->B * r->W;
  if (r->W > 1 && (io->sync_rank < 0 || io->sync_rank >= r->W || !io->sync_hook)) {
    set_error("pn_model: sync_world %d needs 0 <= sync_rank < sync_world and a sync_hook", r->W);
    delete r;
    return PN_ERR_INVALID_ARGUMENT;
  }
  if (r->W > 1 && io->aux_stream) {
    set_error("pn_model: synchronised BatchNormalization runs on one stream (no aux_stream)");
    delete r;
    return PN_ERR_INVALID_ARGUMENT;
  }
  r->prec = d->prec; r->s16 = (d->prec & PN_STORE_BF16) ? 1 : 0; r->st = st; r->training = io->training != 0;
  r->P = io->params; r->G = io->grads;
  r->aux = reinterpret_cast<hipStream_t>(io->aux_stream);
  if (r->aux == st) r->aux = nullptr;
  Arena A;
  A.base = reinterpret_cast<char*>(io->workspace);
  plan_ws(A, r->w, *d, io->B, io->N, r->training);
  if (A.off > io->workspace_bytes) {
    set_error("pn_model: workspace too small (%zu needed, %zu given)", A.off, io->workspace_bytes);
    delete r;
    return PN_ERR_WORKSPACE;
  }
  out = r;
  return PN_OK;
}

}  // namespace pn

using namespace pn;

extern "C" {

int pn_model_num_slots(const pn_model_desc* d) {
  if (check_desc(d) != PN_OK) return -1;
  return (int)make_layout(*d).slots.size();
}
int64_t pn_model_param_floats(const pn_model_desc* d) {
  if (check_desc(d) != PN_OK) return -1;
  return make_layout(*d).total;
}
int pn_model_slot_info(const pn_model_desc* d, int i, pn_slot_info* out) {
  PN_TRY(check_desc(d));
  const Layout L = make_layout(*d);
  PN_CHECK_ARG(out && i >= 0 && i < (int)L.slots.size(), "pn_model_slot_info: index %d out of range", i);
  const Slot& s = L.slots[i];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, s.name.c_str(), sizeof(out->name) - 1);
  out->offset = s.off; out->rows = s.rows; out->cols = s.cols; out->kind = s.kind; out->block = s.block;
  return PN_OK;
}
size_t pn_model_workspace_bytes(const pn_model_desc* d, int B, int N, int training) {
  if (check_desc(d) != PN_OK || B <= 0 || N <= 0) return 0;
  Arena A;
  WS w;
  plan_ws(A, w, *d, B, N, training != 0);
  return A.off;
}
int pn_model_ws_lookup(const pn_model_desc* d, int B, int N, int training, const char* name, int64_t* offset, int64_t* bytes) {
  PN_TRY(check_desc(d));
  PN_CHECK_ARG(name && offset && bytes && B > 0 && N > 0, "pn_model_ws_lookup: bad arguments");
  Arena A;
  WS w;
  std::vector<std::pair<std::string, std::pair<size_t, size_t>>> dir;
  A.dir = &dir;
  plan_ws(A, w, *d, B, N, training != 0);
  for (auto& e : dir)
    if (e.first == name) {
      *offset = (int64_t)e.second.first;
      *bytes = (int64_t)e.second.second;
      return PN_OK;
    }
  set_error("pn_model_ws_lookup: no workspace buffer named '%s'", name);
  return PN_ERR_INVALID_ARGUMENT;
}
int pn_model_ws_entry(const pn_model_desc* d, int B, int N, int training, int index, char* name_out, int name_cap, int64_t* offset,
                      int64_t* bytes) {
  PN_TRY(check_desc(d));
  PN_CHECK_ARG(name_out && name_cap > 0 && offset && bytes && B > 0 && N > 0, "pn_model_ws_entry: bad arguments");
  Arena A;
  WS w;
  std::vector<std::pair<std::string, std::pair<size_t, size_t>>> dir;
  A.dir = &dir;
  plan_ws(A, w, *d, B, N, training != 0);
  if (index < 0 || index >= (int)dir.size()) return 1;   // past the end
  strncpy(name_out, dir[index].first.c_str(), name_cap - 1);
  name_out[name_cap - 1] = 0;
  *offset = (int64_t)dir[index].second.first;
  *bytes = (int64_t)dir[index].second.second;
  return PN_OK;
}
int pn_model_forward(const pn_model_desc* d, const pn_model_io* io, pn_stream stream) {
  Run* r = nullptr;
  PN_TRY(make_run(d, io, reinterpret_cast<hipStream_t>(stream), r));
  const int rc = r->forward();
  delete r;
  return rc;
}
int pn_model_backward(const pn_model_desc* d, const pn_model_io* io, const float* d_cls, const float* d_seg, const float* d_R,
                      pn_stream stream) {
  Run* r = nullptr;
  PN_TRY(make_run(d, io, reinterpret_cast<hipStream_t>(stream), r));
  int rc;
  if (!r->training) {
    set_error("pn_model_backward: the forward pass must have run with training=1");
    rc = PN_ERR_INVALID_ARGUMENT;
  } else {
    rc = r->backward(d_cls, d_seg, d_R);
    const int rj = r->join();            // also on error: never leave the auxiliary stream forked
    if (rc == PN_OK) rc = rj;
  }
  delete r;
  return rc;
}
int pn_adam_prepare(const int32_t* iterations, float* alpha_scratch, double lr0, double decay_rate, double decay_steps, double beta1, double beta2,
                    pn_stream stream) {
  return adam_prepare(iterations, lr0, decay_rate, decay_steps, beta1, beta2, alpha_scratch, reinterpret_cast<hipStream_t>(stream));
}
int pn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int32_t* iterations, float* alpha_scratch, double lr0,
                 double decay_rate, double decay_steps, double beta1, double beta2, double eps, float grad_scale, pn_stream stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  PN_CHECK_ARG(iterations && alpha_scratch, "pn_adam_step: null pointer");
  return adam_fused(params, grads, m, v, n, iterations, lr0, decay_rate, decay_steps, beta1, beta2, eps, grad_scale, alpha_scratch, st);
}

}  // extern "C"
