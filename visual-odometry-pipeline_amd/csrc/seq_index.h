// Which sequence of a group a workgroup serves, and where that sequence's arrays are (see vohip_seqs).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "vohip_internal.h"

namespace vohip_dev {

// Position of the blockIdx.y-th set bit of the group's mask: a few scalar instructions, uniform over the
// workgroup (no table in memory, so no dependent load in front of every kernel of the loop).
__device__ __forceinline__ int seq_index(const vohip_seqs &Q)
{
    unsigned long long m = Q.mask;
    for (unsigned k = blockIdx.y; k > 0; --k) m &= m - 1ull;
    return __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
}

// `p` names an array of sequence 0; the same array of sequence `seq`.
template <class T>
__device__ __forceinline__ T *seq_ptr(T *p, unsigned long long stride, int seq)
{
    using B = typename std::conditional<std::is_const<T>::value, const char, char>::type;
    return reinterpret_cast<T *>(reinterpret_cast<B *>(p) + stride * (unsigned long long)seq);
}

}  // namespace vohip_dev

// Shifts the named pointer variable to the serving sequence's copy (_OPT: a pointer that may be NULL; the
// plain form is an unconditional add, which keeps the pointer's provenance visible to the compiler).
#define VOHIP_SEQ_STATE_OPT(p) \
    if (p) p = vohip_dev::seq_ptr(p, SQ.stride_state, seq__)
#define VOHIP_SEQ_STATE(p) p = vohip_dev::seq_ptr(p, SQ.stride_state, seq__)
#define VOHIP_SEQ_PIN(p) p = vohip_dev::seq_ptr(p, SQ.stride_pin, seq__)
#define VOHIP_SEQ_PREP(p) p = vohip_dev::seq_ptr(p, SQ.stride_prep, seq__)
