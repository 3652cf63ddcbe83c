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
    return 0;
}
