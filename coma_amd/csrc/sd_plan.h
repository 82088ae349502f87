// Internal: recording hook shared by the sd_* launch entry points (see sd_plan.hip).
#pragma once
#include <cstdint>

namespace sd {

enum PlanKind : int { PK_CONV = 1, PK_GN, PK_GN_COLSTATS, PK_LN, PK_ATTN, PK_SOFTMAX, PK_TEMB, PK_COPY, PK_ATTN_WIDE, PK_XCHAIN, PK_XFRONT, PK_GN_TABLE, PK_XTAIL, PK_CONV_SMALL_N, PK_WINO_IN, PK_WINO_OUT, PK_GN_WINO_IN, PK_IM2COL_C3, PK_GN_TABLE_CAT, PK_CONV_HALO, PK_CONV_C3, PK_SEG, PK_COUNT_ };

// One recorded launch: every pointer argument in p[], every integer in i[], every float in f[] (the entry point that records it
// and the replay switch in sd_plan.hip agree on the order).  Pointers are kept apart so that a saved model can be relocated.
struct PlanRec {
  int kind;
  int reserved;
  void* p[16];
  int64_t i[24];
  double f[4];
};

bool plan_recording();                 // is this thread recording into a model?
int plan_record(const PlanRec& r);     // append; returns COMA_OK
int seg_replay(const PlanRec& r, void* stream);      // PK_SEG: i[0] = SEG_OP_* of include/seg_hip.h (seg_ops.hip)

}  // namespace sd
