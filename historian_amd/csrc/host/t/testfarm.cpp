// testfarm: the job-dealing rule of the device farm (lptAssign: longest-processing-time-first list scheduling) on known
// cases.  No device is touched: runs anywhere the libraries load.
#include <cstdio>
#include <cstdlib>
#include "../hx_host.h"
using namespace historian;

static void show(const char* what, const vguard<double>& cost, int devices) {
  const vguard<int> dealt = lptAssign(cost, devices);
  vguard<double> load((size_t)devices, 0.);
  printf("%s:", what);
  for (size_t k = 0; k < cost.size(); ++k) {
    printf(" %d", dealt[k]);
    load[(size_t)dealt[k]] += cost[k];
  }
  printf(" | load");
  for (double l : load) printf(" %g", l);
  printf("\n");
}

int main() {
  show("classic", {7, 7, 6, 6, 5, 4, 4, 2, 2, 2}, 3);          // makespan 16 (optimal 15): the textbook LPT example
  show("equal", {4, 4, 4, 4, 4, 4, 4, 4}, 8);
  show("one-big", {100, 1, 1, 1, 1, 1, 1, 1, 1}, 4);
  show("fewer-jobs", {3, 9}, 8);
  show("single-device", {5, 1, 3}, 1);
  show("tree-level", {4e6, 4e6, 1e6, 9e6, 2.5e5, 6.4e5}, 2);
  return EXIT_SUCCESS;
}
