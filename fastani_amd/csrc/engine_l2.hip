// engine_l2.hip — the two kernels of the L2 stage that do its work, k_l2_codes and k_l2_sim<A / B> (kernels/l2.hpp; ≙
// computeMap.hpp:418-497, slidingMap.hpp, MIIteratorL2.hpp), in a unit of their own so that they can be compiled with their own
// scheduling strategy: csrc/build_lib.sh adds `-mllvm -amdgpu-sched-strategy=max-ilp` for this unit only.  Both kernels are long
// chains of dependent integer instructions per lane with several independent chains side by side; the max-ILP scheduler interleaves
// them (L2 stage 80.7 -> 79.5 ms), while the same flag makes the sketch and L1 kernels 20 - 28 % SLOWER (they live in other units;
// profiles/r05ae_ab_compiler_flags.txt).  engine_map.hip keeps the orchestration and launches through these three functions.
#include "host/engine.hpp"
#include "kernels/l2.hpp"

namespace anih {
using namespace ani;

void launch_l2_codes(unsigned grid, hipStream_t s, const L2FastArgs &fa)
{
  hipLaunchKernelGGL(k_l2_codes, dim3(grid), dim3(kTPB), 0, s, fa);
}
void launch_l2_sim_a(unsigned grid, hipStream_t s, const L2FastArgs &fa, const int32_t *list, const unsigned int *listCount)
{
  hipLaunchKernelGGL((k_l2_sim<L2GeomA>), dim3(grid), dim3(kL2SimTPB), 0, s, fa, list, listCount);
}
void launch_l2_sim_b(unsigned grid, hipStream_t s, const L2FastArgs &fa, const int32_t *list, const unsigned int *listCount)
{
  hipLaunchKernelGGL((k_l2_sim<L2GeomB>), dim3(grid), dim3(kL2SimTPB), 0, s, fa, list, listCount);
}

}  // namespace anih
