// Packed feature-grid descriptor shared by the render kernels and the training path.  Plain C++ (no HIP types) so that
// the host-compiled checks of the per-ray arithmetic (tests/host_math) can use it too.
#ifndef HR_GRID_H
#define HR_GRID_H

// Packed feature grids.  Texel = [density channels | appearance channels] of one
// plane-pair index j, channel counts rounded up to a multiple of 4 floats:
//   static: plane j  [H = N[mat1]][W = N[mat0]][cd4 + ca4], line j [N[vec]][cd4 + ca4]
//   video:  space j  [H][W][cd4 + ca4],                     time j [K][N[matT0]][cd4 + ca4]
struct HrGridPlane {
    const void* a;      // plane (static) / space plane (video): texels of `tex` floats (or halfs, HR_GRID_FP16)
    const void* b;      // line (static) / time plane (video)
    int tex;            // elements per texel: 4*(cd4+ca4), rounded up to a multiple of 8 for halfs (16-byte loads)
    int aw, ah;         // plane width / height in texels
    int bw, bh;         // line: bw = 1, bh = N[vec];  time plane: bw = N[matT0], bh = K
    int cd4, ca4;       // density / appearance float4 groups per texel
    int ax, ay;         // point coordinate index sampled along plane x / y
    int bx;             // point coordinate index sampled along b's axis (line: y; time plane: x)
    int app_off;        // first (padded) appearance slot of this plane pair in the per-ray decode matrix
    int app_real;       // real appearance channels of this plane pair (<= 4*ca4)
    int app_real_off;   // their first column in basis_mat (position in the reference's torch.cat)
};

#endif  // HR_GRID_H
