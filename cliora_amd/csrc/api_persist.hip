// C ABI, persistent unit: instantiations and launch of the level-loop kernels (persist_kernels.hpp).
// A translation unit of its own: the kernels are large and the units compile in parallel.
// Round 5: an OPTIONAL part of the build (-DCLIORA_WITH_PERSISTENT): AUTO selected it for no BASELINE configuration (d = 400: the launches
// win; configs[0]: the sentence-resident kernels come first), so the default library carries only code a default run can reach.
#include "api_common.hpp"
#ifdef CLIORA_WITH_PERSISTENT
#include "level_kernels.hpp"
#include "persist_kernels.hpp"

template <int CT, int K16, bool F32>
static int launch_persist_fwd_inst(hipStream_t st, const PersistFwd& a, int nwg, size_t lds) {
    OKR(cliora_ensure_max_lds((const void*)chart_fwd_persist<CT, K16, F32>));
    hipLaunchKernelGGL((chart_fwd_persist<CT, K16, F32>), dim3(nwg), dim3(512), lds, st, a);
    LAUNCHOK("chart_fwd_persist");
    return CLIORA_OK;
}

// One workgroup per CU; `ct` column tiles per resident weight block (FwdLayout::ct3); the arithmetic mode as the other compose kernels.
int cliora_launch_persist_fwd(hipStream_t st, const PersistFwd& a, int ct, int nwg) {
    const bool f32 = !split_bf16();
    const int S = f32 ? a.Dp : a.S;
    const size_t lds = (size_t)ct * 16 * S * sizeof(uint32_t) + PK_LDS_EXTRA;
    if (lds > 160 * 1024) return fail(CLIORA_EINVAL, "persistent kernel: weight block + scratch exceed LDS");
    PersistFwd b = a;
    b.S = S; b.K = a.Dp;
#define PF_CASE(c, k16) return f32 ? launch_persist_fwd_inst<c, k16, true>(st, b, nwg, lds) : launch_persist_fwd_inst<c, k16, false>(st, b, nwg, lds)
    if (ct == 5 && a.Dp == 400) PF_CASE(5, 25);
    switch (ct) {
        case 5: PF_CASE(5, 0);
        case 4: PF_CASE(4, 0);
        case 2: PF_CASE(2, 0);
        default: PF_CASE(1, 0);
    }
#undef PF_CASE
}
#else
namespace cliora { struct PersistFwd; }
int cliora_launch_persist_fwd(hipStream_t, const cliora::PersistFwd&, int, int) {
    return fail(CLIORA_EINVAL, "the persistent level-loop kernel is not part of this build (-DCLIORA_WITH_PERSISTENT)");
}
#endif
