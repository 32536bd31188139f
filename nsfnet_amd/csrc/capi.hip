// C ABI (include/nsfnet_pinn.h) over the HIP kernels.  Host-side only: argument
// checking, workspace carving and launches.  No device allocation happens here.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>

#include "../../include/nsfnet_pinn.h"
#include "kernels.h"

static_assert((int)PINN_FLD_COUNT == (int)FLD_COUNT, "field plane enum mismatch");

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* a = "", long b = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}
static int hipfail(int rc, const char* what) {
  if (rc == -1000) return fail(-22, "%s: unsupported hidden width (code %ld)", what, (long)rc);
  return fail(rc, "%s: HIP error %ld", what, (long)-rc);
}

// precision of a kernel family: 0 = fp32-input MFMA (exact fp32), 1 = bf16x3 split, 2 = plain bf16
struct pinn_net_s {
  int n_out, L, H, HP;
  int wide;   // 64-column tile kernels (always for HP > 256; PINN_FORCE_WIDE=1 forces them for HP 128/256)
  int prec_fwd, prec_bwd, prec_dw;
};
static int terms_of(int prec) { return prec == 1 ? 3 : 1; }
// 64-column-tile kernels: always for HP > 256; for HP == 256 in fp32 mode they are also the faster
// choice (two workgroups per CU overlap each other's epilogue and MFMA phases): PINN_FORCE_WIDE=0 opts out.
static int env_int(const char* name, int dflt);
static int pick_wide(const pinn_net_s* n) {
  if (n->HP > 256) return 1;
  if (n->HP != 256 && n->HP != 128) return 0;
  const bool fp32 = !n->prec_fwd && !n->prec_bwd && !n->prec_dw;
  // measured (6x256, 360k pts): fp32 26.2 vs 29.6 ms/step in favour of 64-column tiles; bf16x3 13.4 vs 12.6 ms
  // against them.  PINN_TILE_COLS=64|128 overrides the choice (HP 128/256 only).
  const int force = env_int("PINN_TILE_COLS", 0);
  if (force == 64) return 1;
  if (force == 128) return 0;
  return fp32 && n->HP == 256;
}

struct pinn_plan_s {
  pinn_net_s net;
  long n;
  int streams, ntiles, npad;
  int grid_f, grid_b, groups;
  int s24w;              // wide bf16 residual plan (all three kernels bf16): 24-bit three-plane spill format
  int s0_skip;           // the sweeps do not spill layer 0 (role-split pair): dw_bf16 recomputes its activations
  int s0_skip32;         // fp32 residual plan: layer 0 not spilled, recomputed by its readers (FwdArgs::s0_skip)
  int sl0; size_t sblk;  // compact spill geometry of the role-split plans (kernels.h spill_off); sblk = 0: classic layout
  int stagger;           // $PINN_STAGGER, read once at plan creation
  int pipe_f, grid_fp;   // schedule of the forward with saved activations (0 8-wave, 1 pipelined, 2 role-split); grid of 1 / 2 (pairs of tiles)
  int pipe_b;            // schedule of the reverse sweep; for 1 / 2 grid_b is the pair grid
  int wsplit;            // wide net (hidden > 256): role-split sweeps at 64-column tiles (fwd / bwd_bf16_wsplit.hip) instead of the 8-wave ones
  // workspace offsets in bytes
  size_t off_partials, off_oadj, off_sg, off_slabs, off_S, off_Zb, bytes_fwd, bytes_all;
};

// compute units of the CURRENT device (queried per plan: no cached value, a process may drive several devices)
static int num_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
    return n;
  (void)hipGetLastError();
  return 256;   // MI355X (also what a device-less host sizes workspaces for)
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}  // (declared above pick_wide)

// kernel-family dispatch (precision x tile geometry), shared by the launches and by pinn_plan_create's
// configure pass (args.configure = 1: set the dynamic-LDS attribute of exactly the kernel a launch would pick)
static int dispatch_fwd(const pinn_plan_s* plan, const FwdArgs& a, hipStream_t s, bool pipe = false) {
  const pinn_net_s& n = plan->net;
  const int cols = n.wide ? 64 : 128, NS = plan->streams;
  if (pipe && plan->wsplit) return launch_fwd_wsplit(n.HP, terms_of(n.prec_fwd), a, plan->grid_fp, s);
  if (pipe) return plan->pipe_f == 2 ? launch_fwd_split(n.HP, terms_of(n.prec_fwd), a, plan->grid_fp, s)
                                     : launch_fwd_pipe(n.HP, terms_of(n.prec_fwd), a, plan->grid_fp, s);
  if (n.prec_fwd)
    return n.HP > 256 ? launch_fwd_bf16_wide(n.HP, NS, terms_of(n.prec_fwd), a, plan->grid_f, s)
                      : launch_fwd_bf16(n.HP, NS, terms_of(n.prec_fwd), cols, a, plan->grid_f, s);
  return n.wide ? launch_fwd_wide(n.HP, NS, a, plan->grid_f, s) : launch_fwd(n.HP, NS, a, plan->grid_f, s);
}
static int dispatch_bwd(const pinn_plan_s* plan, const BwdArgs& a, hipStream_t s) {
  const pinn_net_s& n = plan->net;
  const int cols = n.wide ? 64 : 128, NS = plan->streams;
  if (plan->wsplit) return launch_bwd_wsplit(n.HP, terms_of(n.prec_bwd), a, plan->grid_b, s);
  if (plan->pipe_b == 2) return launch_bwd_split(n.HP, terms_of(n.prec_bwd), a, plan->grid_b, s);
  if (plan->pipe_b) return launch_bwd_pipe(n.HP, terms_of(n.prec_bwd), a, plan->grid_b, s);
  if (n.prec_bwd)
    return n.HP > 256 ? launch_bwd_bf16_wide(n.HP, NS, terms_of(n.prec_bwd), a, plan->grid_b, s)
                      : launch_bwd_bf16(n.HP, NS, terms_of(n.prec_bwd), cols, a, plan->grid_b, s);
  return n.wide ? launch_bwd_wide(n.HP, NS, a, plan->grid_b, s) : launch_bwd(n.HP, NS, a, plan->grid_b, s);
}
static int dispatch_dw(const pinn_plan_s* plan, const DwArgs& d, hipStream_t s) {
  const pinn_net_s& n = plan->net;
  if (n.prec_dw && n.HP > 256) return launch_dw_bf16_wide(n.HP, plan->streams, terms_of(n.prec_dw), d, s);
  if (n.prec_dw) return launch_dw_bf16(n.HP, plan->streams, terms_of(n.prec_dw), n.wide ? 64 : 128, d, s);
  if (n.wide) return launch_dw_wide(n.HP, plan->streams, d, s);
  return launch_dw(n.HP, plan->streams, d, s);
}

extern "C" {

const char* pinn_last_error(void) { return g_err; }
int pinn_abi_version(void) { return 3; }   // 3: + pinn_plan_kernel; 2: + pinn_adam_step_dev, hidden <= 512 in every precision mode

int pinn_net_create(int n_out, int n_hidden_layers, int hidden, pinn_net_t* out) {
  if (!out) return fail(-22, "pinn_net_create: null out%s");
  if (n_out < 1 || n_out > 3) return fail(-22, "pinn_net_create: n_out must be 1..3%s");
  if (n_hidden_layers < 1 || n_hidden_layers > 64) return fail(-22, "pinn_net_create: hidden layers must be 1..64%s");
  if (hidden < 1 || hidden > PINN_MAX_HP) return fail(-22, "pinn_net_create: hidden width must be 1..512 (got %s%ld)", "", hidden);
  pinn_net_s* n = new (std::nothrow) pinn_net_s;
  if (!n) return fail(-12, "pinn_net_create: out of host memory%s");
  n->n_out = n_out; n->L = n_hidden_layers; n->H = hidden; n->HP = (hidden + 31) / 32 * 32;
  n->prec_fwd = n->prec_bwd = n->prec_dw = 0;
  n->wide = pick_wide(n);
  *out = n;
  return 0;
}
int pinn_net_set_precision(pinn_net_t net, int prec_fwd, int prec_bwd, int prec_dw) {
  if (!net) return fail(-22, "pinn_net_set_precision: null net%s");
  if (prec_fwd < 0 || prec_fwd > 2 || prec_bwd < 0 || prec_bwd > 2 || prec_dw < 0 || prec_dw > 2)
    return fail(-22, "pinn_net_set_precision: precision must be 0 (fp32), 1 (bf16x3) or 2 (bf16)%s");
  // hidden > 256: fwd_bf16_wide / bwd_bf16_wide (64 features per wave) and the blocked dw_bf16_wide
  net->prec_fwd = prec_fwd; net->prec_bwd = prec_bwd; net->prec_dw = prec_dw;
  net->wide = pick_wide(net);
  return 0;
}
int pinn_net_destroy(pinn_net_t net) { delete net; return 0; }
int64_t pinn_net_num_params(pinn_net_t net) { return net ? (int64_t)flat_total(net->H, net->L, net->n_out) : -1; }
int64_t pinn_net_prep_floats(pinn_net_t net) { return net ? (int64_t)prep_total(net->HP, net->L) : -1; }

int pinn_net_prepare(pinn_net_t net, const float* params, float* prep, void* stream) {
  if (!net || !params || !prep) return fail(-22, "pinn_net_prepare: null argument%s");
  int rc = launch_prep(params, prep, net->H, net->HP, net->L, net->n_out, net->prec_fwd, net->prec_bwd,
                       (hipStream_t)stream);
  return rc ? hipfail(rc, "pinn_net_prepare") : 0;
}

int pinn_plan_create(pinn_net_t net, int64_t n_points, int streams, pinn_plan_t* out) {
  if (!net || !out) return fail(-22, "pinn_plan_create: null argument%s");
  if (streams != 1 && streams != 4) return fail(-22, "pinn_plan_create: streams must be 1 or 4%s");
  if (n_points < 1 || n_points > (int64_t)1 << 30) return fail(-22, "pinn_plan_create: bad point count %s%ld", "", (long)n_points);
  pinn_plan_s* p = new (std::nothrow) pinn_plan_s;
  if (!p) return fail(-12, "pinn_plan_create: out of host memory%s");
  p->net = *net;
  p->n = n_points; p->streams = streams;
  const bool wide = net->wide != 0;
  const int per_tile = wide ? (streams == 4 ? 16 : 64) : (streams == 4 ? 32 : 128);
  p->ntiles = (int)((n_points + per_tile - 1) / per_tile);
  p->npad = p->ntiles * per_tile;
  const int HP = net->HP, L = net->L, NW = HP / 32;
  const int cus = num_cus();
  auto bpc = [&](size_t lds) {
    int b = (int)(163840 / lds);
    int bw = NW >= 8 ? (wide && NW == 8 ? 2 : 1) : 8 / NW;
    if (b > bw) b = bw;
    return b < 1 ? 1 : b;
  };
  const int cols = wide ? 64 : 128;
  const bool wbf = HP > 256;   // wide bf16 kernels
  const size_t lds_f = net->prec_fwd ? (wbf ? fwd_bf16_wide_lds_bytes(HP, L) : fwd_bf16_lds_bytes(HP, L, cols)) : wide ? fwd_wide_lds_bytes(HP) : fwd_lds_bytes(HP);
  const size_t lds_b = net->prec_bwd ? (wbf ? bwd_bf16_wide_lds_bytes(HP, L) : bwd_bf16_lds_bytes(HP, L, cols)) : wide ? bwd_wide_lds_bytes(HP, L) : bwd_lds_bytes(HP, L);
  const size_t lds_d = net->prec_dw ? (wbf ? dw_bf16_wide_lds_bytes() : dw_bf16_lds_bytes(HP)) : wide ? dw_wide_lds_bytes() : dw_lds_bytes(HP);
  if (lds_b > 163840 || lds_f > 163840) { delete p; return fail(-22, "pinn_plan_create: this depth x width needs more than 160 KiB of LDS%s"); }
  p->grid_f = cus * bpc(lds_f);
  if (p->grid_f > p->ntiles) p->grid_f = p->ntiles;
  // (forward-only calls, save = 0, always take the 8-wave kernel)
  // Schedule of the hidden-256 bf16 sweeps in residual mode: 0 = 8-wave kernels (fwd_bf16 / bwd_bf16), 1 = one wave
  // per SIMD, two tiles per wave (fwd_bf16_pipe / bwd_bf16_pipe), 2 = two wave groups in opposite phases
  // (fwd_bf16_split / bwd_bf16_split).  $PINN_FWD_SCHED / $PINN_BWD_SCHED choose per sweep, $PINN_SCHED both.
  // Default 2.  Round-2 measurements at 6x256 / 360k points (ms): forward 2.84 / 2.65 / 2.47, reverse sweep
  // 3.75 / 3.40 / 3.22 - the schedules end close to each other because all of them wait on the spill traffic
  // (DESIGN.md section 4.3).
  const bool pipe_shape = HP == 256 && !wide && streams == 4 && L >= 2;
  const int sched_all = env_int("PINN_SCHED", 2);
  int sf = env_int("PINN_FWD_SCHED", sched_all), sb = env_int("PINN_BWD_SCHED", sched_all);
  if (!pipe_shape || !net->prec_fwd) sf = 0;
  if (!pipe_shape || !net->prec_bwd) sb = 0;
  if (sf == 2 && fwd_split_lds_bytes(HP, L) > 163840) sf = 1;
  if (sf == 1 && fwd_pipe_lds_bytes(HP, L) > 163840) sf = 0;
  if (sb == 2 && bwd_split_lds_bytes(HP, L) > 163840) sb = 1;
  if (sb == 1 && bwd_pipe_lds_bytes(HP, L) > 163840) sb = 0;
  p->pipe_f = sf < 0 || sf > 2 ? 0 : sf;
  p->pipe_b = sb < 0 || sb > 2 ? 0 : sb;
  // The role-split sweeps do not spill layer 0 (its saved activations are one FMA pair and one tanh of the point: the
  // reverse sweep and dw_bf16 recompute them), so they only come as a pair, and with the bf16 dW kernel; a request for
  // one of them alone runs that sweep on schedule 1.
  p->s0_skip = p->pipe_f == 2 && p->pipe_b == 2 && net->prec_dw;
  // wide nets (hidden > 256), all three kernels in a bf16 mode, residual mode: the same 24-bit spill format
  p->s24w = HP > 256 && streams == 4 && net->prec_fwd && net->prec_bwd && net->prec_dw;
  if (!p->s0_skip) { if (p->pipe_f == 2) p->pipe_f = 1; if (p->pipe_b == 2) p->pipe_b = 1; }
  // wide nets in the 24-bit format: the role-split sweeps at 64-column tiles where their LDS fits (hidden <= 448: the last
  // K region must fit twice in the 512-element image rows); $PINN_WSPLIT=0 keeps the 8-wave kernels.  Forward-only
  // calls (save = 0) always take the 8-wave kernel.
  p->wsplit = p->s24w && L >= 2 && env_int("PINN_WSPLIT", 1) != 0 && fwd_wsplit_lds_bytes(HP) <= PINN_LDS_MAX &&
              bwd_wsplit_lds_bytes(HP, L) <= PINN_LDS_MAX;
  p->grid_fp = cus < (p->ntiles + 1) / 2 ? cus : (p->ntiles + 1) / 2;
  p->stagger = env_int("PINN_STAGGER", 0);
  if (env_int("PINN_VERBOSE", 0))
    fprintf(stderr, "[pinn] plan: %ld pts, %d streams, HP %d, L %d, prec %d/%d/%d, wide %d, schedule fwd %d bwd %d\n",
            (long)n_points, streams, HP, L, net->prec_fwd, net->prec_bwd, net->prec_dw, (int)wide, p->pipe_f, p->pipe_b);
  p->grid_b = cus * bpc(lds_b);
  if (p->grid_b > p->ntiles) p->grid_b = p->ntiles;
  if (p->pipe_b || p->wsplit) p->grid_b = p->grid_fp;
  if (L > 1) {
    int g = cus * bpc(lds_d) / (L - 1);
    if (g < 1) g = 1;
    if (g > p->ntiles) g = p->ntiles;
    p->groups = g;
  } else {
    p->groups = 0;
  }
  size_t off = 0;
  p->off_partials = off; off = align_up(off + (size_t)(p->grid_f > p->grid_fp ? p->grid_f : p->grid_fp) * PINN_NLOSS * 4, 256);
  p->off_oadj = off;     off = align_up(off + (size_t)4 * p->npad * 4, 256);
  p->bytes_fwd = off;
  p->off_sg = off;       off = align_up(off + (size_t)p->grid_b * sg_total(HP, L) * 4, 256);
  p->off_slabs = off;    off = align_up(off + (size_t)(L - 1) * p->groups * HP * HP * 4, 256);
  const size_t ablk = (size_t)HP * (wide ? 64 : PINN_TILE_COLS);
  // The role-split pair writes three 16-byte planes per register quad and no layer 0: its S and Z-bar are sized for
  // exactly that, (L - 1) blocks of 3/4 of the classic HP x 128 floats per tile (5.5 instead of 8.9 GB each at
  // 6x256 / 360 000 points).  Every other plan keeps the classic [tile][L][HP x columns] layout.
  p->s0_skip32 = streams == 4 && L >= 2 && !net->prec_fwd && !net->prec_bwd && !net->prec_dw && env_int("PINN_S0_SKIP32", 1) != 0;
  p->sl0 = p->s0_skip ? 1 : 0;
  p->sblk = p->s0_skip ? ablk / 4 * 3 : 0;
  const size_t spill_tile = p->s0_skip ? (size_t)(L - 1) * p->sblk : (size_t)L * ablk;      // floats per tile
  // (+1 tile: the pipelined kernels work on PAIRS of tiles; an odd count's dummy partner spills into this scratch block)
  p->off_S = off;        off = align_up(off + (size_t)(p->ntiles + 1) * spill_tile * 4, 256);
  p->off_Zb = off;       off = align_up(off + (size_t)(p->ntiles + 1) * spill_tile * 4, 256);
  p->bytes_all = off;
  // Raise the dynamic-LDS limit of the kernels this plan will launch, on the CURRENT device.  It is per-device
  // state of the HIP runtime and idempotent; doing it here, per plan, keeps the launch path free of cached
  // "already configured" flags (no global mutable state; a process may drive several devices and threads).
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) { ndev = 0; (void)hipGetLastError(); }
  if (ndev > 0) {     // (a host without a device can still size workspaces; it cannot launch anyway)
    FwdArgs fa; memset(&fa, 0, sizeof(fa)); fa.L = L; fa.configure = 1;
    BwdArgs ba; memset(&ba, 0, sizeof(ba)); ba.L = L; ba.configure = 1;
    DwArgs da;  memset(&da, 0, sizeof(da)); da.L = L; da.groups = p->groups; da.configure = 1; da.s0_skip = p->s0_skip; da.s24 = p->s24w;
    int rc = dispatch_fwd(p, fa, nullptr);
    if (!rc && (p->pipe_f || p->wsplit)) rc = dispatch_fwd(p, fa, nullptr, true);
    if (!rc) rc = dispatch_bwd(p, ba, nullptr);
    if (!rc) rc = dispatch_dw(p, da, nullptr);
    if (rc) { delete p; return hipfail(rc, "pinn_plan_create(kernel attributes)"); }
  }
  *out = p;
  return 0;
}
int pinn_plan_destroy(pinn_plan_t plan) { delete plan; return 0; }
int64_t pinn_plan_padded_points(pinn_plan_t plan) { return plan ? plan->npad : -1; }
const char* pinn_plan_kernel(pinn_plan_t plan, int which) {
  if (!plan || which < 0 || which > 2) return nullptr;
  const pinn_net_s& n = plan->net;
  const bool wbf = n.HP > 256;
  if (which == 0 && plan->wsplit) return "fwd_wsplit_kernel";
  if (which == 1 && plan->wsplit) return "bwd_wsplit_kernel";
  if (which == 0) return plan->pipe_f == 2 ? "fwd_split_kernel" : plan->pipe_f ? "fwd_pipe_kernel" : n.prec_fwd ? (wbf ? "fwd_bf16_wide_kernel" : "fwd_bf16_kernel")
                                      : n.wide ? "fwd_wide_kernel" : "fwd_kernel";
  if (which == 1) return plan->pipe_b == 2 ? "bwd_split_kernel" : plan->pipe_b ? "bwd_pipe_kernel" : n.prec_bwd ? (wbf ? "bwd_bf16_wide_kernel" : "bwd_bf16_kernel")
                                      : n.wide ? "bwd_wide_kernel" : "bwd_kernel";
  return n.prec_dw ? (wbf ? "dw_bf16_wide_kernel" : "dw_bf16_kernel") : n.wide ? "dw_wide_kernel" : "dw_kernel";
}
int64_t pinn_plan_workspace_bytes(pinn_plan_t plan, int with_backward) {
  if (!plan) return -1;
  return (int64_t)(with_backward ? plan->bytes_all : plan->bytes_fwd);
}

#define WS(p, off) reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (p)->off)

int pinn_residual_forward(pinn_plan_t plan, void* ws, const float* prep,
                          const float* x, const float* y, const float* e, const float* w,
                          float* vis_t_minus, float* vis_t_out, float* fields,
                          float Re, float vis_t0, float alpha_evm, float coord_scale,
                          int save, float* loss_sums, void* stream) {
  if (!plan || !ws || !prep || !x || !y || !fields) return fail(-22, "pinn_residual_forward: null argument%s");
  if (plan->streams != 4) return fail(-22, "pinn_residual_forward: plan is not a residual (4-stream) plan%s");
  if (!(Re > 0.f)) return fail(-22, "pinn_residual_forward: Re must be > 0%s");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.n = (int)plan->n; a.ntiles = plan->ntiles; a.L = plan->net.L; a.n_out = plan->net.n_out;
  a.prep = prep; a.S = save ? WS(plan, off_S) : nullptr;
  a.fld = fields; a.e = e; a.w = w; a.vtm = vis_t_minus; a.vis_used = vis_t_out;
  a.inv_re = 1.0f / Re; a.vis_t0 = vis_t0; a.alpha_evm = alpha_evm; a.scale = coord_scale;
  a.s24 = plan->s24w;
  a.partials = WS(plan, off_partials);
  a.stagger = plan->ntiles > 4 * plan->grid_f ? plan->stagger : 0;
  a.sl0 = plan->sl0; a.sblk = plan->sblk; a.s0_skip = plan->s0_skip32;
  const bool pipe = (plan->pipe_f || plan->wsplit) && save;
  int rc = dispatch_fwd(plan, a, (hipStream_t)stream, pipe);
  if (rc) return hipfail(rc, "pinn_residual_forward");
  if (loss_sums) {
    rc = launch_loss_sums(a.partials, pipe ? plan->grid_fp : plan->grid_f, loss_sums, (hipStream_t)stream);
    if (rc) return hipfail(rc, "pinn_residual_forward(loss sums)");
  }
  return 0;
}

static int run_dw_and_stash(pinn_plan_t plan, void* ws, const float* prep, const float* x, const float* y, hipStream_t s) {
  DwArgs d;
  memset(&d, 0, sizeof(d));
  d.S = WS(plan, off_S); d.Zb = WS(plan, off_Zb);
  d.ntiles = plan->ntiles; d.L = plan->net.L; d.groups = plan->groups;
  d.slabs = WS(plan, off_slabs);
  d.configure = 0;
  d.s0_skip = plan->s0_skip || plan->s0_skip32; d.s24 = plan->s24w; d.x = x; d.y = y; d.prep = prep; d.n = (int)plan->n;
  d.sl0 = plan->sl0; d.sblk = plan->sblk;
  return dispatch_dw(plan, d, s);
}

int pinn_residual_backward_phases(pinn_plan_t plan, void* ws, const float* prep,
                                  const float* x, const float* y, const float* e, const float* w,
                                  const float* vis_t, const float* fields, const float* coef_eq4,
                                  float Re, float coord_scale, float* ebar_out, int phases, void* stream) {
  if (!plan || !ws || !prep || !x || !y || !fields || !coef_eq4) return fail(-22, "pinn_residual_backward: null argument%s");
  if (plan->streams != 4) return fail(-22, "pinn_residual_backward: plan is not a residual (4-stream) plan%s");
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.n = (int)plan->n; a.ntiles = plan->ntiles; a.L = plan->net.L; a.n_out = plan->net.n_out;
  a.prep = prep; a.S = WS(plan, off_S); a.Zb = WS(plan, off_Zb);
  a.fld = fields; a.e = e; a.w = w; a.vis_used = vis_t;
  for (int k = 0; k < 4; ++k) a.coef_eq[k] = coef_eq4[k];
  a.inv_re = 1.0f / Re; a.scale = coord_scale; a.ebar = ebar_out;
  a.s24 = plan->s24w;
  a.sl0 = plan->sl0; a.sblk = plan->sblk; a.s0_skip = plan->s0_skip32;
  a.sg = WS(plan, off_sg);
  int rc = 0;
  if (phases & 1) {
    rc = dispatch_bwd(plan, a, (hipStream_t)stream);
    if (rc) return hipfail(rc, "pinn_residual_backward");
  }
  if (phases & 2) rc = run_dw_and_stash(plan, ws, prep, x, y, (hipStream_t)stream);
  return rc ? hipfail(rc, "pinn_residual_backward(dW)") : 0;
}

int pinn_residual_backward(pinn_plan_t plan, void* ws, const float* prep,
                           const float* x, const float* y, const float* e, const float* w,
                           const float* vis_t, const float* fields, const float* coef_eq4,
                           float Re, float coord_scale, float* ebar_out, void* stream) {
  return pinn_residual_backward_phases(plan, ws, prep, x, y, e, w, vis_t, fields, coef_eq4, Re, coord_scale,
                                       ebar_out, 3, stream);
}

int pinn_value_forward(pinn_plan_t plan, void* ws, const float* prep,
                       const float* x, const float* y,
                       float* const* pred3, const float* const* tgt3, const float* coef3,
                       int save, float* loss_sums, void* stream) {
  if (!plan || !ws || !prep || !x || !y) return fail(-22, "pinn_value_forward: null argument%s");
  if (plan->streams != 1) return fail(-22, "pinn_value_forward: plan is not a value (1-stream) plan%s");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.n = (int)plan->n; a.ntiles = plan->ntiles; a.L = plan->net.L; a.n_out = plan->net.n_out;
  a.prep = prep; a.S = save ? WS(plan, off_S) : nullptr;
  for (int c = 0; c < 3; ++c) {
    a.pred[c] = (pred3 && c < a.n_out) ? pred3[c] : nullptr;
    a.tgt[c] = (tgt3 && c < a.n_out) ? tgt3[c] : nullptr;
    a.coef[c] = coef3 ? coef3[c] : 0.f;
  }
  a.oadj = save ? WS(plan, off_oadj) : nullptr;
  a.scale = 1.f;
  a.partials = WS(plan, off_partials);
  int rc = dispatch_fwd(plan, a, (hipStream_t)stream);
  if (rc) return hipfail(rc, "pinn_value_forward");
  if (loss_sums) {
    rc = launch_loss_sums(a.partials, plan->grid_f, loss_sums, (hipStream_t)stream);
    if (rc) return hipfail(rc, "pinn_value_forward(loss sums)");
  }
  return 0;
}

int pinn_value_backward(pinn_plan_t plan, void* ws, const float* prep,
                        const float* x, const float* y, const float* out_adj, void* stream) {
  if (!plan || !ws || !prep || !x || !y) return fail(-22, "pinn_value_backward: null argument%s");
  if (plan->streams != 1) return fail(-22, "pinn_value_backward: plan is not a value (1-stream) plan%s");
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.y = y; a.n = (int)plan->n; a.ntiles = plan->ntiles; a.L = plan->net.L; a.n_out = plan->net.n_out;
  a.prep = prep; a.S = WS(plan, off_S); a.Zb = WS(plan, off_Zb);
  a.oadj = out_adj ? out_adj : WS(plan, off_oadj);
  a.scale = 1.f;
  a.sg = WS(plan, off_sg);
  int rc = dispatch_bwd(plan, a, (hipStream_t)stream);
  if (rc) return hipfail(rc, "pinn_value_backward");
  rc = run_dw_and_stash(plan, ws, prep, x, y, (hipStream_t)stream);
  return rc ? hipfail(rc, "pinn_value_backward(dW)") : 0;
}

int pinn_grad_reduce(pinn_net_t net, int nsrc, const pinn_plan_t* plans, void* const* wss,
                     float* grads, int accumulate, void* stream) {
  if (!net || !plans || !wss || !grads) return fail(-22, "pinn_grad_reduce: null argument%s");
  if (nsrc < 1 || nsrc > 4) return fail(-22, "pinn_grad_reduce: nsrc must be 1..4%s");
  ReduceArgs r;
  memset(&r, 0, sizeof(r));
  r.nsrc = nsrc; r.H = net->H; r.HP = net->HP; r.L = net->L; r.n_out = net->n_out;
  r.grads = grads; r.accumulate = accumulate;
  for (int k = 0; k < nsrc; ++k) {
    pinn_plan_t p = plans[k];
    void* ws = wss[k];
    if (!p || !ws) return fail(-22, "pinn_grad_reduce: null plan/workspace%s");
    if (p->net.H != net->H || p->net.L != net->L || p->net.n_out != net->n_out)  // (precision may differ)
      return fail(-22, "pinn_grad_reduce: plan belongs to a different net%s");
    r.src[k].slabs = WS(p, off_slabs); r.src[k].groups = p->groups;
    r.src[k].sg = WS(p, off_sg); r.src[k].nwg = p->grid_b;
  }
  int rc = launch_reduce(r, (hipStream_t)stream);
  return rc ? hipfail(rc, "pinn_grad_reduce") : 0;
}

int pinn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
  if (!params || !grads || !m || !v) return fail(-22, "pinn_adam_step: null argument%s");
  if (step < 1) return fail(-22, "pinn_adam_step: step must be >= 1%s");
  double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  int rc = launch_adam(params, grads, m, v, (long)n, (float)((double)lr / bc1), beta1, beta2, eps, (float)std::sqrt(bc2),
                       (hipStream_t)stream);
  return rc ? hipfail(rc, "pinn_adam_step") : 0;
}

int pinn_adam_step_dev(float* params, const float* grads, float* m, float* v, int64_t n,
                       float lr, float beta1, float beta2, float eps, int64_t* step_counter, void* stream) {
  if (!params || !grads || !m || !v || !step_counter) return fail(-22, "pinn_adam_step_dev: null argument%s");
  int rc = launch_adam_dev(params, grads, m, v, (long)n, lr, beta1, beta2, eps,
                           reinterpret_cast<long long*>(step_counter), (hipStream_t)stream);
  return rc ? hipfail(rc, "pinn_adam_step_dev") : 0;
}

}  // extern "C"
