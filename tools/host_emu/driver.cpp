// Runs one problem file (written by tools/host_emu/dump_problem.py) through the kernel compiled for the host.
#define ALILQR_SINGLE_TU 1
#include "../../ilqr_amd/csrc/alilqr_api.hip"
#include <vector>
namespace alilqr { double lds_pool[20480]; }   // 160 KB: the LDS of one CU

static std::vector<double> rd(FILE* f, size_t n) { std::vector<double> v(n); if (n && fread(v.data(), 8, n, f) != n) { perror("read"); exit(2); } return v; }

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  alilqr_problem p;
  if (fread(&p, sizeof(p), 1, f) != 1) return 2;
  const size_t B = p.batch, H = p.horizon, n = p.n, m = p.m, c = p.c;
  const size_t cw = (p.mode == ALILQR_MODE_LEGACY) ? 2 * (n + m) : c * (n + m + 1) + ((p.cons_kind & ALILQR_CONS_QUADRATIC) ? c * n * n : 0);
  auto x0 = rd(f, B * n), goals = rd(f, p.goal_batch * (H + 1) * n), Q = rd(f, p.table_nodes * n * n),
       R = rd(f, p.table_nodes * m * m), cons = rd(f, p.table_nodes * cw);
  // optional trailer: n_left, n_right, first_row, pad (int32) + rectangles
  int32_t tr[4] = {0, 0, 0, 0};
  std::vector<double> lobs, robs;
  alilqr_extras ex{};
  if (fread(tr, 4, 4, f) == 4) {
    lobs = rd(f, (size_t)tr[0] * 8); robs = rd(f, (size_t)tr[1] * 8);
    ex.left_obs = lobs.data(); ex.right_obs = robs.data(); ex.n_left = tr[0]; ex.n_right = tr[1]; ex.first_row = tr[2];
  }
  std::vector<double> x(B * (H + 1) * n), u(B * H * m), nx(x.size()), nu(u.size()), xi(x.size()), ui(u.size()),
      cost(B), viol(B), lam(B * (H + 1) * c), mu(B);
  std::vector<int32_t> st(B * 4);
  // the emulated group width (threads per group, ALILQR_LANES) must be the kernel's lane-group width
  alilqr_set_tuning(emu_group_width(), 0, 0);
  int rc = alilqr_solve_batch_ex(&p, x0.data(), goals.data(), Q.data(), R.data(), cons.data(), &ex, x.data(), u.data(), nx.data(), nu.data(),
                              xi.data(), ui.data(), cost.data(), viol.data(), lam.data(), mu.data(), st.data(), nullptr);
  printf("rc %d %s\n", rc, alilqr_last_error());
  for (size_t b = 0; b < B; ++b) printf("[%zu] cost %.12g viol %.3e mu %g status %d %d %d %d\n", b, cost[b], viol[b], mu[b], st[4 * b], st[4 * b + 1], st[4 * b + 2], st[4 * b + 3]);
  if (argc > 2) {
    FILE* o = fopen(argv[2], "wb");
    fwrite(x.data(), 8, x.size(), o); fwrite(u.data(), 8, u.size(), o); fwrite(cost.data(), 8, B, o); fwrite(st.data(), 4, st.size(), o);
    fclose(o);
  }
  alilqr_release();
  return 0;
}
