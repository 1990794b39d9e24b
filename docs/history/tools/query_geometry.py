import ctypes, torch
hip = ctypes.CDLL(torch.__file__.rsplit("/",1)[0] + "/lib/libamdhip64.so")
v = ctypes.c_int(0)
# enum positions: find hipDeviceAttributeNumberOfXccs by scanning the header
import re
src = open("/opt/rocm/include/hip/hip_runtime_api.h").read()
body = src[src.index("typedef enum hipDeviceAttribute_t"):]
body = body[:body.index("} hipDeviceAttribute_t")]
names = [m.group(1) for m in re.finditer(r"^\s*(hipDeviceAttribute\w+)\s*(=\s*[^,]+)?,", body, re.M)]
print(len(names))
vals = {}
cur = -1
for m in re.finditer(r"^\s*(hipDeviceAttribute\w+)\s*(?:=\s*([^,/]+))?,", body, re.M):
    n, e = m.group(1), m.group(2)
    if e:
        e = e.strip()
        cur = vals[e] if e in vals else int(e, 0)
    else:
        cur += 1
    vals[n] = cur
for n in ("hipDeviceAttributeMultiprocessorCount", "hipDeviceAttributeNumberOfXccs"):
    r = hip.hipDeviceGetAttribute(ctypes.byref(v), vals[n], 0)
    print(n, vals[n], "rc", r, "value", v.value)
print(torch.cuda.get_device_properties(0))
