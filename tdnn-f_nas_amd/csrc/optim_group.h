// optim_group.h -- the trainer's per-minibatch optimizer step as a handful of grouped launches (optim_group.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace tdnnf {

struct UpdComp {
  long long begin, end;  // [begin, end) in the flat buffers, alignment padding included; both multiples of 4
  int rows, cols;        // the weight matrix at `begin` (row-major, ld = cols)
  float orthonormal;     // 0: not constrained (nnet-utils.cc:1047-1061)
};
struct UpdGroup;

// params: the flat parameter buffer the task lists point into (the group is rebuilt by the caller when it moves)
int upd_group_create(const std::vector<UpdComp> &comps, float *params, UpdGroup **out);
void upd_group_destroy(UpdGroup *g);
const float *upd_group_params(const UpdGroup *g);
// delta = lr_c * grads + l2coef_c * params (written into grads), per-component and global max-change (UpdateNnetWithMaxChange with
// scale = max_change_scale = 1), params += factor_c * delta, grads = 0: three launches.
int upd_group_step(UpdGroup *g, float *params, float *grads, const float *lr, const float *l2coef, const float *max_change, float max_param_change,
                   hipStream_t s);
// ConstrainOrthonormalInternal for the selected components (indices into `comps`; each must be constrained and have rows <= cols),
// all of them together in five launches.
int upd_group_ortho(UpdGroup *g, const std::vector<int> &selected, hipStream_t s);
bool upd_group_can_ortho(const UpdGroup *g, int comp);

}  // namespace tdnnf
