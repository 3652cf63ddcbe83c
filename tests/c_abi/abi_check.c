/* TEST-ONLY: include/hyperreel_hip.h must be plain C and the library must be usable without Python.
 * Built by tests/test_host_logic.py with `gcc -std=c99 -pedantic`; exercises only calls that need no GPU. */
#include <stdio.h>
#include <string.h>

#include "hyperreel_hip.h"

int main(void)
{
    hr_config cfg;
    hr_model* m = NULL;
    int rc;
    memset(&cfg, 0, sizeof(cfg));
    printf("abi %d sizeof %d %d\n", hr_abi_version(), hr_sizeof_config(), (int)sizeof(hr_config));
    if (hr_abi_version() != HR_ABI_VERSION) return 1;
    if (hr_sizeof_config() != (int)sizeof(hr_config)) return 2;      /* same layout under a C and a C++ compiler */
    rc = hr_model_create(NULL, &m);
    if (rc != HR_E_INVALID || m != NULL) return 3;
    rc = hr_model_create(&cfg, &m);                                   /* all-zero config: rejected with a message */
    if (rc != HR_E_INVALID || m != NULL || strlen(hr_last_error()) == 0) return 4;
    printf("error text: %s\n", hr_last_error());
    hr_model_destroy(NULL);                                           /* no-op by contract */
    if (hr_render(NULL, NULL, 0, NULL, NULL) != HR_E_INVALID) return 5;
    if (hr_render_frame(NULL, NULL, 0, 0.5f, NULL, NULL) != HR_E_INVALID) return 5;
    /* every other entry point refuses bad arguments before it touches the device */
    {
        hr_train_tensors t;
        int32_t n3[3] = {2, 2, 2};
        float box[6] = {0, 0, 0, 1, 1, 1}, x = 0.0f;
        unsigned char px[4];
        memset(&t, 0, sizeof(t));
        if (hr_model_update_config(NULL, &cfg, NULL) != HR_E_INVALID) return 6;
        if (hr_train_features(NULL, NULL, 0, NULL, NULL) != HR_E_INVALID) return 7;
        if (hr_train_forward(NULL, &t, NULL, NULL, 0, 0, NULL, NULL) != HR_E_INVALID) return 8;
        if (hr_train_backward(NULL, NULL, NULL, NULL, 0, 0, NULL, &t, NULL) != HR_E_INVALID) return 9;
        if (hr_plane_reg_forward(&x, 1, 0, 4, &x, NULL) != HR_E_INVALID) return 10;         /* h < 1 */
        if (hr_plane_reg_backward(&x, 1, 4, 4, NULL, &x, NULL) != HR_E_INVALID) return 11;  /* no coefficients */
        if (hr_upsample_plane(&x, 1, 0, 1, &x, 1, 1, NULL) != HR_E_INVALID) return 12;
        if (hr_pack_display(&x, 0, 4, 0, 0, 1, px, NULL) != HR_E_INVALID) return 13;
        if (hr_pack_display(NULL, 4, 4, 0, 0, 1, px, NULL) != HR_E_INVALID) return 14;
        if (hr_dense_alpha(NULL, n3, 0.01f, 1, NULL, NULL, box, &x, NULL) != HR_E_INVALID) return 15;
        if (hr_generate_rays(NULL, 6, 0, 0, NULL, NULL) != HR_E_INVALID) return 16;
        if (hr_model_create_cascade(NULL, &cfg, &m) != HR_E_INVALID || m != NULL) return 17;
        if (hr_model_set_option(NULL, HR_OPT_FRAME_KERNEL, 1) != HR_E_INVALID) return 18;
        if (hr_linear_forward(&x, 4, 8, 4, &x, NULL, 0, 0.01f, &x, 4, NULL) != HR_E_INVALID) return 20;          /* out < 1 */
        if (hr_linear_backward(&x, 4, &x, NULL, 4, &x, 4, 8, 4, 4, 0.01f, NULL, 4, &x, &x, &x, NULL) != HR_E_INVALID) return 21;   /* mask missing */
        if (hr_linear_workspace(0, 4, 4) != 0) return 22;
        if (hr_model_set_occupancy(NULL, NULL, n3, box, NULL) != HR_E_INVALID) return 23;
        { int32_t v = 0; if (hr_model_get_option(NULL, HR_OPT_SAMPLE_WAVES, &v) != HR_E_INVALID) return 19; }
        if (hr_model_calibrate(NULL, &x, 1, NULL, NULL) != HR_E_INVALID) return 24;
        { hr_verify_info vi; if (hr_model_verify_info(NULL, &vi) != HR_E_INVALID || sizeof(hr_verify_info) != 96) return 32; }
        if (hr_allgather_tiles(NULL, &x, &x, 3, NULL) != HR_E_INVALID) return 25;
        {   /* hr_adam_step: NULL arrays refused; a negative size and a step count of 0 refused before anything is launched; zero tensors is a no-op */
            float* pp[1] = {&x}; const float* gp[1] = {&x}; int64_t nn[1] = {-1}; double hp[6] = {1e-3, 0.9, 0.99, 1e-8, 0.0, 1.0};
            if (hr_adam_step(NULL, NULL, NULL, NULL, NULL, NULL, 1, NULL) != HR_E_INVALID) return 28;
            if (hr_adam_step(pp, gp, pp, pp, nn, hp, 1, NULL) != HR_E_INVALID) return 29;
            nn[0] = 4; hp[5] = 0.0;
            if (hr_adam_step(pp, gp, pp, pp, nn, hp, 1, NULL) != HR_E_INVALID) return 30;
            if (hr_adam_step(NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) != HR_OK) return 31;
        }
        { int64_t f = -1, c = -1; if (hr_shard_range(10, 1, 4, &f, &c) != HR_OK || f != 3 || c != 3) return 26; if (hr_shard_range(10, 4, 4, &f, &c) != HR_E_INVALID) return 27; }
    }
    return 0;
}
