// TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.
//
// Thin extern "C" driver over the reference's own gco-v3.0 C++ library
// (GCoptimizationGeneralGraph, /root/reference/gco_source/GCoptimization.h:551-597),
// compiled IN PLACE from /root/reference/gco_source by oracle/Makefile into
// oracle/_ref/libgco_ref.so.  No gco source is copied into this repository; this
// file only *calls* the library's public API (setDataCost / setSmoothCost /
// setNeighbors / setLabel / swap / expansion / whatLabel / compute_energy), the
// same calls the un-vendored `pygco` wrapper issues underneath
// `pygco.cut_general_graph` (reference call site phylo_hmrf.py:496-498).
//
// The handle-based shape follows what a ctypes wrapper needs: create, set the three
// cost arrays, set labels, run swap/expansion, read labels/energies, destroy.
// Errors: gco throws GCException (GCoptimization.cpp:1072-1076); it is caught here
// and turned into a non-zero status + message (no exception crosses the C boundary).

#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "GCoptimization.h"

namespace {
typedef GCoptimizationGeneralGraph GG;
std::map<int, GG*> g_graphs;
int g_next = 1;
std::string g_err;

GG* find(int h) {
  std::map<int, GG*>::iterator it = g_graphs.find(h);
  return it == g_graphs.end() ? 0 : it->second;
}
}  // namespace

#define GCOREF_TRY(h) \
  GG* gc = find(h);   \
  if (!gc) {          \
    g_err = "bad handle"; \
    return 2;         \
  }                   \
  try {
#define GCOREF_END                 \
  }                                \
  catch (GCException e) {          \
    g_err = e.message ? e.message : "GCException"; \
    return 1;                      \
  }                                \
  return 0;

extern "C" {

const char* gcoref_last_error() { return g_err.c_str(); }

int gcoref_max_energy_term() { return GCO_MAX_ENERGYTERM; }

int gcoref_create_general_graph(int n_sites, int n_labels, int* handle) {
  try {
    GG* gc = new GG(n_sites, n_labels);
    g_graphs[g_next] = gc;
    *handle = g_next++;
  } catch (GCException e) {
    g_err = e.message ? e.message : "GCException";
    return 1;
  }
  return 0;
}

int gcoref_destroy(int h) {
  GG* gc = find(h);
  if (!gc) return 2;
  delete gc;
  g_graphs.erase(h);
  return 0;
}

// unary: [n_sites * n_labels], site-major (GCoptimization.h:216, setDataCost(array)).
// gco keeps the POINTER (no copy), so the caller must keep the array alive.
int gcoref_set_data_cost(int h, int* unary) {
  GCOREF_TRY(h)
  gc->setDataCost(unary);
  GCOREF_END
}

// smooth: [n_labels * n_labels]; gco keeps the pointer.
int gcoref_set_smooth_cost(int h, int* smooth) {
  GCOREF_TRY(h)
  gc->setSmoothCost(smooth);
  GCOREF_END
}

// One setNeighbors call per undirected pair (GCoptimization.h:563).
int gcoref_set_all_neighbors(int h, const int* s1, const int* s2, const int* w, int n_pairs) {
  GCOREF_TRY(h)
  for (int i = 0; i < n_pairs; ++i) gc->setNeighbors(s1[i], s2[i], w[i]);
  GCOREF_END
}

int gcoref_set_labels(int h, const int* labels, int n) {
  GCOREF_TRY(h)
  for (int i = 0; i < n; ++i) gc->setLabel(i, labels[i]);
  GCOREF_END
}

int gcoref_get_labels(int h, int* labels, int n) {
  GCOREF_TRY(h)
  gc->whatLabel(0, n, labels);
  GCOREF_END
}

int gcoref_swap(int h, int max_iter, long long* energy) {
  GCOREF_TRY(h)
  *energy = gc->swap(max_iter);
  GCOREF_END
}

int gcoref_expansion(int h, int max_iter, long long* energy) {
  GCOREF_TRY(h)
  *energy = gc->expansion(max_iter);
  GCOREF_END
}

int gcoref_alpha_beta_swap(int h, int a, int b) {
  GCOREF_TRY(h)
  gc->alpha_beta_swap(a, b);
  GCOREF_END
}

int gcoref_energy(int h, long long* total, long long* data, long long* smooth) {
  GCOREF_TRY(h)
  *total = gc->compute_energy();
  *data = gc->giveDataEnergy();
  *smooth = gc->giveSmoothEnergy();
  GCOREF_END
}

}  // extern "C"
