// cdf_out.hip -- the output and restart files of the reference WITHOUT PnetCDF (host code; nothing here runs
// on the GPU except the device-to-host copies).  io_pnetcdf.F writes NetCDF "64-bit offset" files
// (nf_64bit_offset = CDF-2) through the parallel library: write_output_pnetcdf (:57-410) and
// write_restart_pnetcdf (:1661-2083).  The classic format is simple enough to emit directly: a header (dimensions,
// global attributes, variables with their attributes and byte offsets) followed by every variable's values as
// big-endian doubles in definition order.  Same dimension names and lengths, same variable names, order,
// dimensions (Fortran's (x,y,zz,time) is (time,zz,y,x) in the file), types (all nf_double, def_var_pnetcdf :6-40)
// and attribute texts, so existing post-processing keeps working.  Every rank writes its own (im,jm) patch at
// (i_global(1), j_global(1)) with pwrite into the one file, as the collective put_vara calls do; rank 0 creates
// it (create = 1), the others open it afterwards (create = 0; the caller orders the two with a barrier).
// One deliberate difference: the reference passes length 26 for the 10-character text of vtot's
// formula_terms attribute (:133-135, reading past the literal); here the attribute is the 10 characters.
#include <fcntl.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#ifndef POMGPU_EMU
#include <thread>
#endif

#include "pomgpu.h"
#include "pomgpu_internal.hpp"
#define fail pomgpu_fail

namespace {
struct Att { std::string name, text; };
enum Src { SCALAR, LEVELS1D, PLANE2D, VOLUME3D };
struct Var {
  std::string name;
  std::vector<int> dims;          // dimension ids, slowest first (file order)
  std::vector<Att> atts;
  Src src; int slot; int nlev;    // slot: blk1d / blk2d / blk3d member; nlev: levels written (3-D), values (1-D)
  double value;                   // SCALAR
  uint64_t begin;
};
struct Spec { std::vector<std::pair<std::string, int>> dims; std::vector<Att> gatts; std::vector<Var> vars; };

void put32(std::string &h, uint32_t v) { for (int s = 24; s >= 0; s -= 8) h.push_back((char)((v >> s) & 0xff)); }
void put64(std::string &h, uint64_t v) { for (int s = 56; s >= 0; s -= 8) h.push_back((char)((v >> s) & 0xff)); }
void putname(std::string &h, const std::string &s) { put32(h, (uint32_t)s.size()); h += s; while (h.size() % 4) h.push_back('\0'); }
void putatts(std::string &h, const std::vector<Att> &a) {
  if (a.empty()) { put32(h, 0); put32(h, 0); return; }          // ABSENT
  put32(h, 0x0C); put32(h, (uint32_t)a.size());                  // NC_ATTRIBUTE
  for (const Att &t : a) { putname(h, t.name); put32(h, 2); put32(h, (uint32_t)t.text.size()); h += t.text; while (h.size() % 4) h.push_back('\0'); }
}
uint64_t var_bytes(const Spec &S, const Var &v) { uint64_t n = 8; for (int d : v.dims) n *= (uint64_t)S.dims[d].second; return n; }
std::string header(Spec &S) {
  for (int pass = 0; pass < 2; pass++) {                         // pass 0 measures, pass 1 has the offsets
    std::string h("CDF\002", 4);
    put32(h, 0);                                                 // numrecs: no record dimension
    put32(h, 0x0A); put32(h, (uint32_t)S.dims.size());           // NC_DIMENSION
    for (auto &d : S.dims) { putname(h, d.first); put32(h, (uint32_t)d.second); }
    putatts(h, S.gatts);
    put32(h, 0x0B); put32(h, (uint32_t)S.vars.size());           // NC_VARIABLE
    for (Var &v : S.vars) {
      putname(h, v.name); put32(h, (uint32_t)v.dims.size());
      for (int d : v.dims) put32(h, (uint32_t)d);
      putatts(h, v.atts);
      put32(h, 6);                                               // NC_DOUBLE
      const uint64_t nb = var_bytes(S, v);
      put32(h, nb > 0xffffffffULL ? 0xffffffffu : (uint32_t)nb); // vsize (saturates, as the format prescribes)
      put64(h, v.begin);
    }
    if (pass == 1) return h;
    uint64_t off = h.size();
    for (Var &v : S.vars) { v.begin = off; off += var_bytes(S, v); }
  }
  return std::string();
}
Var mk(const char *name, std::vector<int> dims, const char *long_name, const char *units, const char *coords, Src src, int slot, int nlev,
       double value = 0.) {
  Var v; v.name = name; v.dims = dims; v.src = src; v.slot = slot; v.nlev = nlev; v.value = value; v.begin = 0;
  v.atts.push_back({"long_name", long_name}); v.atts.push_back({"units", units});
  if (coords) v.atts.push_back({"coordinates", coords});
  return v;
}
int write_be(int fd, const double *x, size_t n, uint64_t off, std::vector<uint64_t> &tmp) {
  tmp.resize(n);
  for (size_t q = 0; q < n; q++) { uint64_t u; memcpy(&u, &x[q], 8); tmp[q] = __builtin_bswap64(u); }
  const char *p = (const char *)tmp.data();
  size_t left = n * 8;
  while (left) { const ssize_t w = pwrite(fd, p, left, (off_t)off); if (w <= 0) return -1; p += w; off += (uint64_t)w; left -= (size_t)w; }
  return 0;
}
}  // namespace

// Writing a file does not hold the model up ("async D2H on a copy stream").  The call (i) lays the file out
// (header, full length) at once, so that other ranks may open it after the caller's barrier, (ii) takes a SNAPSHOT of
// every array the file holds with device-to-device copies on the kernels' stream (a few ms at HBM speed; the model may
// overwrite the arrays right after), records an event and returns; (iii) a host thread waits for that event on a copy
// stream of its own, brings the snapshot over through a pinned buffer, swaps bytes and pwrites.  pomgpu_io_wait() joins
// it (also called by pomgpu_sync, the next write and pomgpu_destroy) and reports its I/O status.  POMGPU_IO_SYNC=1, a
// snapshot that does not fit, and the host emulation keep the old synchronous path.
struct IoJob {
  Spec S;
  std::string path;
  int fd = -1, im = 0, jm = 0, iml = 0, jml = 0, kb = 0, i0 = 1, j0 = 1, im_global = 0, jm_global = 0, create = 0, device = 0;
  size_t n2 = 0;
  double *snap = NULL;                 // device: the snapshot, variables back to back in the order of S.vars
  std::vector<double> b1;              // blk1d (host copy taken at the call)
#ifndef POMGPU_EMU
  std::thread th;
  hipEvent_t ev = NULL;
  hipStream_t st = NULL;
#endif
  int rc = 0, active = 0;
};
static size_t var_doubles(const IoJob &J, const Var &v) { return v.src == PLANE2D ? J.n2 : (v.src == VOLUME3D ? (size_t)v.nlev * J.n2 : 0); }
// rows of one variable (host copy `host`, leading dimensions iml x jml) into the file
static int put_var(const IoJob &J, const Var &v, const double *host, std::vector<uint64_t> &tmp) {
  const int nlev = v.src == PLANE2D ? 1 : v.nlev;
  int bad = 0;
  for (int k = 0; k < nlev && !bad; k++)
    for (int j = 0; j < J.jm && !bad; j++) {
      const uint64_t cell = ((uint64_t)k * J.jm_global + (uint64_t)(J.j0 - 1 + j)) * J.im_global + (uint64_t)(J.i0 - 1);
      bad |= write_be(J.fd, host + ((size_t)k * J.jml + j) * J.iml, (size_t)J.im, v.begin + cell * 8, tmp);
    }
  return bad;
}
#ifndef POMGPU_EMU
static void io_worker(IoJob *J) {
  (void)hipSetDevice(J->device);
  int bad = 0;
  size_t big = 0;
  for (const Var &v : J->S.vars) { const size_t n = var_doubles(*J, v); if (n > big) big = n; }
  double *pin = NULL;
  if (big && hipHostMalloc((void **)&pin, big * sizeof(double), hipHostMallocDefault) != hipSuccess) bad = 1;
  if (!bad && hipStreamWaitEvent(J->st, J->ev, 0) != hipSuccess) bad = 1;
  std::vector<uint64_t> tmp;
  size_t off = 0;
  for (const Var &v : J->S.vars) {
    const size_t n = var_doubles(*J, v);
    if (!n) continue;
    if (!bad && (hipMemcpyAsync(pin, J->snap + off, n * sizeof(double), hipMemcpyDeviceToHost, J->st) != hipSuccess ||
                 hipStreamSynchronize(J->st) != hipSuccess)) bad = 1;
    if (!bad) bad |= put_var(*J, v, pin, tmp);
    off += n;
  }
  if (pin) (void)hipHostFree(pin);
  if (close(J->fd)) bad = 1;
  J->fd = -1;
  J->rc = bad;
}
#endif
extern "C" int pomgpu_io_wait(pomgpu_ctx *c) {
  if (!c) return POMGPU_EINVAL;
  IoJob *J = (IoJob *)c->io_job;
  if (!J) return POMGPU_OK;
  int rc = POMGPU_OK;
#ifndef POMGPU_EMU
  if (J->active) {
    J->th.join();
    if (J->rc) rc = fail(c, POMGPU_EINVAL, "write: I/O error on %s", J->path.c_str());
  }
  (void)hipFree(J->snap);
  if (J->ev) (void)hipEventDestroy(J->ev);
  if (J->st) (void)hipStreamDestroy(J->st);
#endif
  delete J;
  c->io_job = NULL;
  return rc;
}

static int write_file(pomgpu_ctx *c, const char *path, const pomgpu_file_meta *m, Spec &S) {
  const KP &P = c->P;
  if (m->i0 < 1 || m->j0 < 1 || m->i0 + P.im - 1 > m->im_global || m->j0 + P.jm - 1 > m->jm_global)
    return fail(c, POMGPU_EINVAL, "write: the tile (%d..%d, %d..%d) does not fit the global grid %d x %d", m->i0, m->i0 + P.im - 1, m->j0,
                m->j0 + P.jm - 1, m->im_global, m->jm_global);
  { const int rcw = pomgpu_io_wait(c); if (rcw) return rcw; }  // one file in flight at a time
  // the snapshot below reads the mirrors directly: whatever the library keeps lazily is brought up to date first, whether or not
  // the caller supplied the statistics (pomgpu_domain_stats would have done it as a side effect)
  { const int rcm = pomgpu_materialize(c); if (rcm) return rcm; }
  const std::string h = header(S);
  const int fd = open(path, m->create ? (O_WRONLY | O_CREAT | O_TRUNC) : O_WRONLY, 0644);
  if (fd < 0) return fail(c, POMGPU_EINVAL, "write: cannot open %s", path);
  std::vector<uint64_t> tmp;
  std::vector<double> host;
  int bad = 0;
  if (m->create) {
    size_t left = h.size(); const char *p = h.data(); off_t off = 0;
    while (left && !bad) { const ssize_t w = pwrite(fd, p, left, off); if (w <= 0) bad = 1; else { p += w; off += w; left -= (size_t)w; } }
    const Var &last = S.vars.back();                                   // full length even where no tile has written yet
    if (!bad && ftruncate(fd, (off_t)(last.begin + var_bytes(S, last)))) bad = 1;
  }
  IoJob *J = new IoJob();
  J->path = path; J->fd = fd; J->im = P.im; J->jm = P.jm; J->iml = P.iml; J->jml = P.jml; J->kb = P.kb; J->n2 = P.n2;
  J->i0 = m->i0; J->j0 = m->j0; J->im_global = m->im_global; J->jm_global = m->jm_global; J->create = m->create; J->device = c->device;
  // scalars and the vertical grid: small, written here
  J->b1.resize((size_t)POM_NBLK1D * P.kb);
  if (hipMemcpyAsync(J->b1.data(), P.b1, sizeof(double) * J->b1.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) bad = 1;
  for (const Var &v : S.vars) {
    if (bad) break;
    if (v.src == SCALAR) { if (m->create) bad |= write_be(fd, &v.value, 1, v.begin, tmp); continue; }
    if (v.src == LEVELS1D && m->create) bad |= write_be(fd, J->b1.data() + (size_t)v.slot * P.kb, (size_t)v.nlev, v.begin, tmp);
  }
  size_t total = 0;
  for (const Var &v : S.vars) total += var_doubles(*J, v);
  bool async = !bad && total > 0 && !SW(c, IO_SYNC);
#ifdef POMGPU_EMU
  async = false;
#else
  if (async && hipMalloc((void **)&J->snap, total * sizeof(double)) != hipSuccess) { J->snap = NULL; (void)hipGetLastError(); async = false; }
#endif
  auto dev_of = [&](const Var &v) { return v.src == PLANE2D ? P.b2 + (size_t)v.slot * P.n2 : P.b3 + (size_t)v.slot * P.a3; };
  if (!async) {                                                        // the synchronous path: variable by variable through pageable memory
    for (const Var &v : S.vars) {
      const size_t n = var_doubles(*J, v);
      if (bad || !n) continue;
      host.resize(n);
      if (hipMemcpyAsync(host.data(), dev_of(v), sizeof(double) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
          hipStreamSynchronize(c->stream) != hipSuccess) { bad = 1; break; }
      bad |= put_var(*J, v, host.data(), tmp);
    }
    if (close(fd)) bad = 1;
    J->fd = -1;
    c->io_job = J;
    (void)pomgpu_io_wait(c);
    return bad ? fail(c, POMGPU_EINVAL, "write: I/O error on %s", path) : POMGPU_OK;
  }
#ifndef POMGPU_EMU
  size_t off = 0;
  for (const Var &v : S.vars) {                                        // the snapshot, on the kernels' stream
    const size_t n = var_doubles(*J, v);
    if (!n) continue;
    if (hipMemcpyAsync(J->snap + off, dev_of(v), n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) bad = 1;
    off += n;
  }
  if (hipEventCreateWithFlags(&J->ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(J->ev, c->stream) != hipSuccess ||
      hipStreamCreateWithFlags(&J->st, hipStreamNonBlocking) != hipSuccess) bad = 1;
  J->S = S;
  c->io_job = J;
  if (bad) { (void)close(fd); J->fd = -1; (void)pomgpu_io_wait(c); return fail(c, POMGPU_EHIP, "write: cannot start the copy of %s", path); }
  J->active = 1;
  J->th = std::thread(io_worker, J);
#endif
  return POMGPU_OK;
}

static void stats_for_file(pomgpu_ctx *c, const pomgpu_file_meta *m, double *s8) {
  if (m->stats) { memcpy(s8, m->stats, 8 * sizeof(double)); return; }   // the caller's rank-reduced values
  (void)pomgpu_domain_stats(c, s8, 0);
}

extern "C" int pomgpu_write_output(pomgpu_ctx *c, const char *path, const pomgpu_file_meta *m) {   // io_pnetcdf.F:57-410
#ifdef POMGPU_STORE_F32
  if (c) return pomgpu_fail(c, POMGPU_EINVAL, "write_output: not in the fp32-storage variant (download and write from the host)");
#endif
  if (!c || !path || !m) return POMGPU_EINVAL;
  (void)hipSetDevice(c->device);
  const KP &P = c->P;
  double s8[8];
  stats_for_file(c, m, s8);                                   // also brings every lazily kept array up to date (NEED)
  Spec S;
  S.dims = {{"time", 1}, {"z", P.kb}, {"zz", P.kbm1}, {"y", m->jm_global}, {"x", m->im_global}};
  S.gatts = {{"title", m->title ? m->title : ""}, {"description", "output file"}};
  const std::string since = std::string("days since ") + (m->time_start ? m->time_start : "");
  const int T = 0, Z = 1, ZZ = 2, Y = 3, X = 4;
  S.vars.push_back(mk("time", {T}, "time", since.c_str(), NULL, SCALAR, 0, 0, c->con.time));
  S.vars.push_back(mk("vtot", {T}, "domain total volume", "metre^3", NULL, SCALAR, 0, 0, s8[0]));
  S.vars.back().atts.push_back({"standard_name", "basin total volume"});
  S.vars.back().atts.push_back({"formula_terms", "time: time"});
  S.vars.push_back(mk("mtot", {T}, "domain total mass", "kg^3", NULL, SCALAR, 0, 0, s8[2]));
  S.vars.push_back(mk("tavg", {T}, "domain average temperature", "degrees Celsius", NULL, SCALAR, 0, 0, s8[4]));
  S.vars.push_back(mk("savg", {T}, "domain average salinity", "psu", NULL, SCALAR, 0, 0, s8[5]));
  S.vars.push_back(mk("eavg", {T}, "domain potential energy (anomaly)", "metre", NULL, SCALAR, 0, 0, s8[6]));
  S.vars.push_back(mk("ekin", {T}, "domain kinetic energy", "J", NULL, SCALAR, 0, 0, s8[7]));
  S.vars.push_back(mk("z", {Z}, "sigma of cell face", "sigma_level", NULL, LEVELS1D, P1_z, P.kb));
  S.vars.back().atts.push_back({"standard_name", "ocean_sigma_coordinate"});
  S.vars.back().atts.push_back({"formula_terms", "sigma: z eta: elb depth: h"});
  S.vars.push_back(mk("zz", {ZZ}, "sigma of cell centre", "sigma_level", NULL, LEVELS1D, P1_zz, P.kbm1));
  S.vars.back().atts.push_back({"standard_name", "ocean_sigma_coordinate"});
  S.vars.back().atts.push_back({"formula_terms", "sigma: zz eta: elb depth: h"});
  struct { const char *n, *ln, *u, *co; int slot; } p2[] = {
      {"dx", "grid increment in x", "metre", "east_e north_e", P2_dx}, {"dy", "grid increment in y", "metre", "east_e north_e", P2_dy},
      {"east_u", "easting of u-points", "metre", "east_u north_u", P2_east_u}, {"east_v", "easting of v-points", "metre", "east_v north_v", P2_east_v},
      {"east_e", "easting of elevation points", "metre", "east_e north_e", P2_east_e}, {"east_c", "easting of cell corners", "metre", "east_c north_c", P2_east_c},
      {"north_u", "northing of u-points", "metre", "east_u north_u", P2_north_u}, {"north_v", "northing of v-points", "metre", "east_v north_v", P2_north_v},
      {"north_e", "northing of elevation points", "metre", "east_e north_e", P2_north_e}, {"north_c", "northing of cell corners", "metre", "east_c north_c", P2_north_c},
      {"rot", "Rotation angle of x-axis wrt. east", "degree", "east_e north_e", P2_rot}, {"h", "undisturbed water depth", "metre", "east_e north_e", P2_h},
      {"fsm", "free surface mask", "dimensionless", "east_e north_e", P2_fsm}, {"dum", "u-velocity mask", "dimensionless", "east_u north_u", P2_dum},
      {"dvm", "v-velocity mask", "dimensionless", "east_v north_v", P2_dvm}};
  for (auto &q : p2) S.vars.push_back(mk(q.n, {Y, X}, q.ln, q.u, q.co, PLANE2D, q.slot, 1));
  S.vars.push_back(mk("uab", {T, Y, X}, "depth-averaged u", "metre/sec", "east_u north_u", PLANE2D, P2_uab, 1));
  S.vars.push_back(mk("vab", {T, Y, X}, "depth-averaged v", "metre/sec", "east_v north_v", PLANE2D, P2_vab, 1));
  S.vars.push_back(mk("elb", {T, Y, X}, "surface elevation", "metre", "east_e north_e", PLANE2D, P2_elb, 1));
  S.vars.push_back(mk("u", {T, ZZ, Y, X}, "x-velocity", "metre/sec", "east_u north_u zz", VOLUME3D, P3_u, P.kbm1));
  S.vars.push_back(mk("v", {T, ZZ, Y, X}, "y-velocity", "metre/sec", "east_v north_v zz", VOLUME3D, P3_v, P.kbm1));
  S.vars.push_back(mk("t", {T, ZZ, Y, X}, "potential temperature", "K", "east_e north_e zz", VOLUME3D, P3_t, P.kbm1));
  S.vars.push_back(mk("s", {T, ZZ, Y, X}, "salinity x rho / rhoref", "PSS", "east_e north_e zz", VOLUME3D, P3_s, P.kbm1));
  S.vars.push_back(mk("rho", {T, ZZ, Y, X}, "(density-1000)/rhoref", "dimensionless", "east_e north_e zz", VOLUME3D, P3_rho, P.kbm1));
  S.vars.push_back(mk("w", {T, Z, Y, X}, "z-velocity", "metre/sec", "east_e north_e z", VOLUME3D, P3_w, P.kb));
  return write_file(c, path, m, S);
}

extern "C" int pomgpu_write_restart(pomgpu_ctx *c, const char *path, const pomgpu_file_meta *m) {   // io_pnetcdf.F:1661-2083
#ifdef POMGPU_STORE_F32
  if (c) return pomgpu_fail(c, POMGPU_EINVAL, "write_restart: not in the fp32-storage variant (download and write from the host)");
#endif
  if (!c || !path || !m) return POMGPU_EINVAL;
  (void)hipSetDevice(c->device);
  const KP &P = c->P;
  double s8[8];
  stats_for_file(c, m, s8);                                   // (values unused: the call brings the state up to date)
  Spec S;
  S.dims = {{"time", 1}, {"z", P.kb}, {"y", m->jm_global}, {"x", m->im_global}};
  S.gatts = {{"title", m->title ? m->title : ""}, {"description", "restart file"}};
  const std::string since = std::string("days since ") + (m->time_start ? m->time_start : "");
  const int T = 0, Z = 1, Y = 2, X = 3;
  S.vars.push_back(mk("iint", {}, "i_internal", "model internal step number", NULL, SCALAR, 0, 0, (double)c->con.iint));
  S.vars.push_back(mk("time", {T}, "time", since.c_str(), NULL, SCALAR, 0, 0, c->con.time));
  struct { const char *n, *ln, *u, *co; int slot; } p2[] = {
      {"wubot", "x-momentum flux at the bottom", "metre^2/sec^2", "east_u north_u", P2_wubot},
      {"wvbot", "y-momentum flux at the bottom", "metre^2/sec^2", "east_v north_v", P2_wvbot},
      {"aam2d", "vertical average of aam", "metre^2/sec", "east_e north_e", P2_aam2d},
      {"ua", "vertical mean of u", "metre/sec", "east_u north_u", P2_ua}, {"uab", "vertical mean of u at time -dt", "metre/sec", "east_u north_u", P2_uab},
      {"va", "vertical mean of v", "metre/sec", "east_v north_v", P2_va}, {"vab", "vertical mean of v at time -dt", "metre/sec", "east_v north_v", P2_vab},
      {"el", "surface elevation in external mode", "metre", "east_e north_e", P2_el},
      {"elb", "surface elevation in external mode at -dt", "metre", "east_e north_e", P2_elb},
      {"et", "surface elevation in internal mode", "metre", "east_e north_e", P2_et},
      {"etb", "surface elevation in internal mode at -dt", "metre", "east_e north_e", P2_etb},
      {"egb", "surface elevation for pres. grad. at -dt", "metre", "east_e north_e", P2_egb},
      {"utb", "ua time averaged over dti", "metre/sec", "east_u north_u", P2_utb}, {"vtb", "va time averaged over dti", "metre/sec", "east_v north_v", P2_vtb},
      {"adx2d", "vertical integral of advx", "-", "east_u north_u", P2_adx2d}, {"ady2d", "vertical integral of advy", "-", "east_v north_v", P2_ady2d},
      {"advua", "sum of 2nd, 3rd and 4th terms in eq (18)", "-", "east_u north_u", P2_advua},
      {"advva", "sum of 2nd, 3rd and 4th terms in eq (19)", "-", "east_v north_v", P2_advva}};
  for (auto &q : p2) S.vars.push_back(mk(q.n, {Y, X}, q.ln, q.u, q.co, PLANE2D, q.slot, 1));
  struct { const char *n, *ln, *u, *co; int slot; } p3[] = {
      {"u", "x-velocity", "metre/sec", "east_u north_u zz", P3_u}, {"ub", "x-velocity at time -dt", "metre/sec", "east_u north_u zz", P3_ub},
      {"v", "y-velocity", "metre/sec", "east_v north_v zz", P3_v}, {"vb", "y-velocity at time -dt", "metre/sec", "east_v north_v zz", P3_vb},
      {"w", "sigma-velocity", "metre/sec", "east_e north_e zz", P3_w}, {"t", "potential temperature", "K", "east_e north_e zz", P3_t},
      {"tb", "potential temperature at time -dt", "K", "east_e north_e zz", P3_tb}, {"s", "salinity x rho / rhoref", "PSS", "east_e north_e zz", P3_s},
      {"sb", "salinity x rho / rhoref at time -dt", "PSS", "east_e north_e zz", P3_sb},
      {"rho", "(density-1000)/rhoref", "dimensionless", "east_e north_e zz", P3_rho},
      {"km", "vertical kinematic viscosity", "metre^2/sec", "east_e north_e zz", P3_km}, {"kh", "vertical diffusivity", "metre^2/sec", "east_e north_e zz", P3_kh},
      {"kq", "kq", "metre^2/sec", "east_e north_e zz", P3_kq}, {"l", "turbulence length scale", "-", "east_e north_e zz", P3_l},
      {"q2", "twice the turbulent kinetic energy", "metre^2/sec^2", "east_e north_e zz", P3_q2},
      {"q2b", "twice the turbulent kinetic energy at -dt", "metre^2/sec^2", "east_e north_e zz", P3_q2b},
      {"aam", "horizontal kinematic viscosity", "metre^2/sec", "east_e north_e zz", P3_aam}, {"q2l", "q2 x l", "metre^3/sec^2", "east_e north_e zz", P3_q2l},
      {"q2lb", "q2 x l at time -dt", "metre^3/sec^2", "east_e north_e zz", P3_q2lb}};
  for (auto &q : p3) S.vars.push_back(mk(q.n, {Z, Y, X}, q.ln, q.u, q.co, VOLUME3D, q.slot, P.kb));
  return write_file(c, path, m, S);
}
