/* shard_client.c -- ONE rank of a pose graph sharded over ranks, driven from plain C: no Python, no torch (test infrastructure and
 * the model for a Rust host: INTEGRATION.md section 3b is this loop).
 *
 * The staged Gauss-Newton iteration of include/rr_pgo.h ("sharding"): stage 0, ALL-GATHER of exchange buffer 0, stage 1, sum
 * ALL-REDUCE of exchange buffer 1, with the two collectives issued by the caller on the handle's own stream.  Here the caller is
 * this program and the collectives are RCCL's (librccl.so, found with dlopen: the library itself does not link it), on a
 * communicator of WORLD_SIZE ranks; loop control as PoseGraph::optimize (reference src/mapping/pose_graph_optimization.rs:247-303:
 * tolerance 1e-4 on |dx|, errors = 1 + iterations entries).  The reference has no counterpart for the sharding itself (its only
 * parallel construct is rayon in update_nodes, :230).
 *
 *   shard_client <file.g2o | grid:WxH[:E]> <precision f64|f32|mixed> <iterations> <out.bin> [id-file]
 *     RANK / WORLD_SIZE / LOCAL_RANK from the environment (default 0 / 1 / RANK): one process per GPU.  WORLD_SIZE > 1: rank 0
 *     writes the RCCL unique id to <id-file>, the others wait for it (a shared file system stands in for the launcher's
 *     rendezvous).  out.bin (rank 0... every rank writes <out.bin>.<rank>): int32 n_errors, int32 state_len, the errors, the |dx|
 *     per iteration (n_errors - 1), the rank's copy of the state -- compared bit for bit with rustrobotics_amd.sharding's drivers
 *     by tests/test_gpu_parity.py.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "rr_pgo.h"

/* the slice of rccl.h this program needs (RCCL = NCCL's API) */
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
enum { NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_SUM = 0 };
typedef int (*ncclGetUniqueId_t)(ncclUniqueId *);
typedef int (*ncclCommInitRank_t)(ncclComm_t *, int, ncclUniqueId, int);
typedef int (*ncclAllGather_t)(const void *, void *, size_t, int, ncclComm_t, void *);
typedef int (*ncclAllReduce_t)(const void *, void *, size_t, int, int, ncclComm_t, void *);
typedef int (*ncclCommDestroy_t)(ncclComm_t);
typedef const char *(*ncclGetErrorString_t)(int);
typedef int (*hipSetDevice_t)(int);

static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
static int fail(const char *what, int rc) {
  fprintf(stderr, "shard_client: %s failed with %d: %s\n", what, rc, rr_pgo_last_error());
  return 2;
}
static void *must_sym(void *lib, const char *name) {
  void *p = dlsym(lib, name);
  if (!p) { fprintf(stderr, "shard_client: %s not found: %s\n", name, dlerror()); exit(3); }
  return p;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: shard_client <file.g2o | grid:WxH[:E]> <f64|f32|mixed> <iterations> <out.bin> [id-file]\n");
    return 1;
  }
  const char *what = argv[1], *prec = argv[2], *out_path = argv[4], *id_path = argc > 5 ? argv[5] : NULL;
  const int iterations = atoi(argv[3]);
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
  int rc;

  /* ---- RCCL, and the device of this rank (before the library creates its stream) */
  void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  void *rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!rccl) rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!hip || !rccl) { fprintf(stderr, "shard_client: dlopen: %s\n", dlerror()); return 3; }
  hipSetDevice_t p_hipSetDevice = (hipSetDevice_t)must_sym(hip, "hipSetDevice");
  ncclGetUniqueId_t p_getid = (ncclGetUniqueId_t)must_sym(rccl, "ncclGetUniqueId");
  ncclCommInitRank_t p_init = (ncclCommInitRank_t)must_sym(rccl, "ncclCommInitRank");
  ncclAllGather_t p_allgather = (ncclAllGather_t)must_sym(rccl, "ncclAllGather");
  ncclAllReduce_t p_allreduce = (ncclAllReduce_t)must_sym(rccl, "ncclAllReduce");
  ncclCommDestroy_t p_destroy = (ncclCommDestroy_t)must_sym(rccl, "ncclCommDestroy");
  ncclGetErrorString_t p_errstr = (ncclGetErrorString_t)must_sym(rccl, "ncclGetErrorString");
  if (p_hipSetDevice(local) != 0) { fprintf(stderr, "shard_client: hipSetDevice(%d) failed\n", local); return 3; }

  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  if (rank == 0) {
    if ((rc = p_getid(&id)) != 0) { fprintf(stderr, "shard_client: ncclGetUniqueId: %s\n", p_errstr(rc)); return 3; }
    if (world > 1) {
      char tmp[4096];
      FILE *f;
      if (!id_path) { fprintf(stderr, "shard_client: WORLD_SIZE > 1 needs an id-file\n"); return 1; }
      snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
      f = fopen(tmp, "wb");
      if (!f || fwrite(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "shard_client: cannot write %s\n", tmp); return 3; }
      fclose(f);
      rename(tmp, id_path);   /* appears complete or not at all */
    }
  } else {
    int tries = 0;
    FILE *f = NULL;
    if (!id_path) { fprintf(stderr, "shard_client: WORLD_SIZE > 1 needs an id-file\n"); return 1; }
    while (!(f = fopen(id_path, "rb")) && tries++ < 600) usleep(100000);
    if (!f || fread(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "shard_client: no unique id in %s\n", id_path); return 3; }
    fclose(f);
  }
  ncclComm_t comm = NULL;
  if ((rc = p_init(&comm, world, id, rank)) != 0) { fprintf(stderr, "shard_client: ncclCommInitRank: %s\n", p_errstr(rc)); return 3; }

  /* ---- the rank's handle: every rank builds it from the SAME graph */
  rr_pgo_options opt;
  rr_pgo *h = NULL;
  rr_pgo_synth *synth = NULL;
  rr_pgo_default_options(&opt);
  opt.precision = strcmp(prec, "f32") == 0 ? RR_PGO_F32 : strcmp(prec, "mixed") == 0 ? RR_PGO_MIXED : RR_PGO_F64;
  opt.device = local;
  opt.rank = rank;
  opt.world_size = world;
  opt.sharded = 1;
  if (strncmp(what, "grid:", 5) == 0) {
    int w = 0, hgt = 0, e = 0;
    rr_pgo_graph_desc d;
    if (sscanf(what + 5, "%dx%d:%d", &w, &hgt, &e) < 2) { fprintf(stderr, "shard_client: bad grid spec\n"); return 1; }
    if ((rc = rr_pgo_synth_grid(w, hgt, e, 42, 43, &synth, &d)) != RR_PGO_OK) return fail("rr_pgo_synth_grid", rc);
    rc = rr_pgo_create(&d, &opt, &h);
  } else {
    rc = rr_pgo_load_g2o(what, &opt, &h);
  }
  if (rc != RR_PGO_OK) return fail("create", rc);
  if (synth) rr_pgo_synth_free(synth);

  void *xch = NULL, *scal = NULL, *stream = rr_pgo_stream(h);
  int64_t xn = 0, sn = 0;
  int32_t xes = 0, ses = 0;
  if ((rc = rr_pgo_exchange_buffer(h, 0, &xch, &xn, &xes)) != RR_PGO_OK) return fail("rr_pgo_exchange_buffer(0)", rc);
  if ((rc = rr_pgo_exchange_buffer(h, 1, &scal, &sn, &ses)) != RR_PGO_OK) return fail("rr_pgo_exchange_buffer(1)", rc);
  const int64_t chunk = xn / world;
  const int xtype = xes == 8 ? NCCL_FLOAT64 : NCCL_FLOAT32;

  /* ---- optimize (:247-303), Gauss-Newton branch, over the two stages and the two collectives */
  double *errors = (double *)calloc((size_t)iterations + 1, sizeof(double)), *norms = (double *)calloc((size_t)iterations + 1, sizeof(double));
  int n_errors = 0, it;
  for (it = 0; it < iterations; it++) {
    double chi, nrm;
    if ((rc = rr_pgo_stage(h, 0, 0.0, 0)) != RR_PGO_OK) return fail("rr_pgo_stage(0)", rc);
    /* in place: rank r contributes elements [r * chunk, (r + 1) * chunk) */
    if ((rc = p_allgather((const char *)xch + (size_t)rank * (size_t)chunk * (size_t)xes, xch, (size_t)chunk, xtype, comm, stream)) != 0) {
      fprintf(stderr, "shard_client: ncclAllGather: %s\n", p_errstr(rc));
      return 3;
    }
    if ((rc = rr_pgo_stage(h, 1, 0.0, 0)) != RR_PGO_OK) return fail("rr_pgo_stage(1)", rc);
    if ((rc = p_allreduce(scal, scal, 2, NCCL_FLOAT64, NCCL_SUM, comm, stream)) != 0) {
      fprintf(stderr, "shard_client: ncclAllReduce: %s\n", p_errstr(rc));
      return 3;
    }
    if ((rc = rr_pgo_stage_scalars(h, &chi, &nrm)) != RR_PGO_OK) return fail("rr_pgo_stage_scalars", rc);   /* synchronises the stream */
    errors[n_errors] = chi;
    norms[n_errors++] = nrm;
    if (nrm < 1e-4) break; /* :298-300 */
  }
  {
    double chi, nrm;
    if ((rc = rr_pgo_stage(h, 2, 0.0, 0)) != RR_PGO_OK) return fail("rr_pgo_stage(2)", rc);
    if ((rc = p_allreduce(scal, scal, 2, NCCL_FLOAT64, NCCL_SUM, comm, stream)) != 0) return 3;
    if ((rc = rr_pgo_stage_scalars(h, &chi, &nrm)) != RR_PGO_OK) return fail("rr_pgo_stage_scalars", rc);
    errors[n_errors++] = chi;
  }

  const int state_len = rr_pgo_state_len(h);
  double *state = (double *)malloc(sizeof(double) * (size_t)state_len);
  if ((rc = rr_pgo_get_state(h, state)) != RR_PGO_OK) return fail("rr_pgo_get_state", rc);
  {
    char path[4096];
    FILE *f;
    int32_t hdr[2];
    snprintf(path, sizeof path, "%s.%d", out_path, rank);
    f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "shard_client: cannot write %s\n", path); return 3; }
    hdr[0] = n_errors; hdr[1] = state_len;
    fwrite(hdr, sizeof hdr, 1, f);
    fwrite(errors, sizeof(double), (size_t)n_errors, f);
    fwrite(norms, sizeof(double), (size_t)(n_errors - 1), f);
    fwrite(state, sizeof(double), (size_t)state_len, f);
    fclose(f);
  }
  if (rank == 0)
    printf("shard_client: %d rank(s), %d iterations, chi2 %.9g -> %.9g, all-gather %lld x %d bytes per rank and iteration\n", world, n_errors - 1, errors[0],
           errors[n_errors - 1], (long long)chunk, (int)xes);
  rr_pgo_destroy(h);
  p_destroy(comm);
  free(errors); free(norms); free(state);
  return 0;
}
