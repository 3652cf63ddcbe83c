// TEST INFRASTRUCTURE (tests/test_gpu_poison.py; never linked into the product library).
//
// hr_poison(pattern, stream): leaves `pattern` in EVERY vector register and EVERY LDS word of every CU, on `stream`, so that the next
// kernel on that stream starts on known garbage.  A kernel that reads a register or an LDS word before writing it computes from whatever the
// previous occupant of the CU left there: alone on a stream that is always the same kernel -- the same garbage, a reproducible image, a
// latent bug; beside another launch it is that launch's data, and the image changes (one ray of a frame, rarely, when two
// models render on two streams).  Rendering the same rays after different patterns and comparing every word turns the rare event into a
// deterministic one.
//
// Coverage on gfx950 (MI355X_MICROARCH.md): a SIMD's register file is 512 VGPRs per lane, handed out in blocks of 8, at most 8 wavefronts
// per SIMD -- 8 x 64.  The kernel is compiled for EXACTLY 64 VGPRs (amdgpu_num_vgpr; no AGPRs) and 512 threads with 40 KB of LDS per block:
// four blocks fill a CU's 32 wavefront slots and its 160 KB of LDS at once.  Every wavefront writes v0..v63 and then sleeps ~60 us, longer
// than the grid takes to be placed, so that all slots are occupied simultaneously; 2 x (4 x CUs) blocks are launched so that a CU that was
// skipped in the first wave of placements gets blocks from the second.
#include <hip/hip_runtime.h>

extern "C" __global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(64), amdgpu_waves_per_eu(8, 8)))
void hr_poison_kernel(unsigned pattern, unsigned* touched)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += 512) lds[i] = pattern;
    __syncthreads();
    if (touched && threadIdx.x == 0) atomicAdd(touched, 1u);          // blocks that ran (the test reads it)
    unsigned cnt;
    asm volatile("s_mov_b32 %0, 8\n"
                 "v_mov_b32 v0, %1\n"
                 "v_mov_b32 v1, %1\n"
                 "v_mov_b32 v2, %1\n"
                 "v_mov_b32 v3, %1\n"
                 "v_mov_b32 v4, %1\n"
                 "v_mov_b32 v5, %1\n"
                 "v_mov_b32 v6, %1\n"
                 "v_mov_b32 v7, %1\n"
                 "v_mov_b32 v8, %1\n"
                 "v_mov_b32 v9, %1\n"
                 "v_mov_b32 v10, %1\n"
                 "v_mov_b32 v11, %1\n"
                 "v_mov_b32 v12, %1\n"
                 "v_mov_b32 v13, %1\n"
                 "v_mov_b32 v14, %1\n"
                 "v_mov_b32 v15, %1\n"
                 "v_mov_b32 v16, %1\n"
                 "v_mov_b32 v17, %1\n"
                 "v_mov_b32 v18, %1\n"
                 "v_mov_b32 v19, %1\n"
                 "v_mov_b32 v20, %1\n"
                 "v_mov_b32 v21, %1\n"
                 "v_mov_b32 v22, %1\n"
                 "v_mov_b32 v23, %1\n"
                 "v_mov_b32 v24, %1\n"
                 "v_mov_b32 v25, %1\n"
                 "v_mov_b32 v26, %1\n"
                 "v_mov_b32 v27, %1\n"
                 "v_mov_b32 v28, %1\n"
                 "v_mov_b32 v29, %1\n"
                 "v_mov_b32 v30, %1\n"
                 "v_mov_b32 v31, %1\n"
                 "v_mov_b32 v32, %1\n"
                 "v_mov_b32 v33, %1\n"
                 "v_mov_b32 v34, %1\n"
                 "v_mov_b32 v35, %1\n"
                 "v_mov_b32 v36, %1\n"
                 "v_mov_b32 v37, %1\n"
                 "v_mov_b32 v38, %1\n"
                 "v_mov_b32 v39, %1\n"
                 "v_mov_b32 v40, %1\n"
                 "v_mov_b32 v41, %1\n"
                 "v_mov_b32 v42, %1\n"
                 "v_mov_b32 v43, %1\n"
                 "v_mov_b32 v44, %1\n"
                 "v_mov_b32 v45, %1\n"
                 "v_mov_b32 v46, %1\n"
                 "v_mov_b32 v47, %1\n"
                 "v_mov_b32 v48, %1\n"
                 "v_mov_b32 v49, %1\n"
                 "v_mov_b32 v50, %1\n"
                 "v_mov_b32 v51, %1\n"
                 "v_mov_b32 v52, %1\n"
                 "v_mov_b32 v53, %1\n"
                 "v_mov_b32 v54, %1\n"
                 "v_mov_b32 v55, %1\n"
                 "v_mov_b32 v56, %1\n"
                 "v_mov_b32 v57, %1\n"
                 "v_mov_b32 v58, %1\n"
                 "v_mov_b32 v59, %1\n"
                 "v_mov_b32 v60, %1\n"
                 "v_mov_b32 v61, %1\n"
                 "v_mov_b32 v62, %1\n"
                 "v_mov_b32 v63, %1\n"
                 "L_hr_poison_%=:\n"
                 "s_sleep 127\n"
                 "s_sub_u32 %0, %0, 1\n"
                 "s_cmp_lg_u32 %0, 0\n"
                 "s_cbranch_scc1 L_hr_poison_%=\n"
                 : "=&s"(cnt) : "s"(pattern)
                 : "scc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

extern "C" int hr_poison(unsigned pattern, unsigned* touched_dev, void* stream)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    hipLaunchKernelGGL(hr_poison_kernel, dim3(8 * cus), dim3(512), 40 * 1024, (hipStream_t)stream, pattern, touched_dev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
