/* zjni_forward.c — trampolines for the natives the GPU path has no business with (context streams, pledged size / progression, training).
 *
 * zstd-jni loads one library, so this one must export all 149 symbols of the reference's.  Each symbol
 * listed in forward_list.h (generated: the reference's exports minus what zjni_shim.c defines) is a signature-agnostic
 * tail jump into the bundled CPU library's function of the same name — argument registers and stack are untouched, so no
 * prototype is restated here and no reference code is linked.  The targets are resolved once, when this library is loaded
 * ($ZSTD_JNI_CPU_LIB); without the bundled library a trampoline returns 0 / NULL.  x86-64 System V only (what the image and
 * the GPU box are).  nativePtr of the context classes IS the bundled library's handle (zjni_shim.c), so pointers pass
 * through unchanged. */
#include <stddef.h>
void* zjni_shim_cpu_sym(const char* name);

#define FWD(name) \
    void* zjni_slot_##name __attribute__((visibility("hidden"))); \
    __attribute__((naked, visibility("default"))) void Java_com_github_luben_zstd_##name(void) { \
        __asm__("movq zjni_slot_" #name "(%rip), %rax\n\ttestq %rax, %rax\n\tjz 1f\n\tjmp *%rax\n1:\txorl %eax, %eax\n\tret"); }
#include "forward_list.h"
#undef FWD

static const struct { const char* name; void** slot; } g_fwd[] = {
#define FWD(name) { "Java_com_github_luben_zstd_" #name, &zjni_slot_##name },
#include "forward_list.h"
#undef FWD
    { NULL, NULL }
};
__attribute__((constructor)) static void zjni_forward_resolve(void) {
    for (size_t i = 0; g_fwd[i].name; i++) *g_fwd[i].slot = zjni_shim_cpu_sym(g_fwd[i].name);
}
