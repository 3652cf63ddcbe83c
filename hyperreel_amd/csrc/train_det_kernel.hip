// Deterministic build of the sample stage's training kernels (HR_OPT_TRAIN_DETERMINISTIC): the same source as train_kernel.hip with
// every gradient accumulator -- texel gradients in HBM, the per-ray decode-matrix gradient and basis_mat's in LDS -- a 64-bit
// fixed-point integer (hr_train.h: hr_acc_t; the unit is chosen per step and kept per model) added with integer atomics.  Integer addition is associative: whatever
// order the memory system retires the adds in, two runs of a step produce the same bits.  The reference's training loop
// (INRSystem.training_step, nlf/__init__.py:634-709) is deterministic for a given thread count; this is the mode that gives the same
// guarantee.  Slower than the default build (no LDS windows for the contended lines / time-plane rows, 8-byte atomics).
#define HR_TRAIN_DET 1
#include <cstring>
#include "train_kernel.hip"
