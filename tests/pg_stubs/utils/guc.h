/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_GUC_H
#define CRYO_STUB_GUC_H
typedef enum { PGC_INTERNAL, PGC_POSTMASTER, PGC_SIGHUP, PGC_SU_BACKEND, PGC_BACKEND, PGC_SUSET, PGC_USERSET } GucContext;
typedef int GucSource;
struct config_enum_entry { const char *name; int val; bool hidden; };
typedef bool (*GucIntCheckHook)(int *newval, void **extra, GucSource source);
typedef void (*GucIntAssignHook)(int newval, void *extra);
typedef bool (*GucEnumCheckHook)(int *newval, void **extra, GucSource source);
typedef void (*GucEnumAssignHook)(int newval, void *extra);
typedef const char *(*GucShowHook)(void);
extern void DefineCustomIntVariable(const char *name, const char *short_desc, const char *long_desc, int *valueAddr, int bootValue,
                                    int minValue, int maxValue, GucContext context, int flags, GucIntCheckHook check_hook,
                                    GucIntAssignHook assign_hook, GucShowHook show_hook);
extern void DefineCustomEnumVariable(const char *name, const char *short_desc, const char *long_desc, int *valueAddr, int bootValue,
                                     const struct config_enum_entry *options, GucContext context, int flags,
                                     GucEnumCheckHook check_hook, GucEnumAssignHook assign_hook, GucShowHook show_hook);
#endif
