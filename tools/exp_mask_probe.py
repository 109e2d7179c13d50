import ctypes as C, torch
torch.zeros(1).cuda()
hip = C.CDLL("libamdhip64.so")
def try_mask(bits, nwords=8):
    words = (C.c_uint32 * nwords)()
    for b in bits: words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(nwords), words)
    if rc == 0: hip.hipStreamDestroy(st)
    return rc
for name, bits in [("bit0", [0]), ("bits0-1", [0, 1]), ("bits0-3", range(4)), ("bits0-7", range(8)), ("bits0-15", range(16)),
                   ("bits0-31", range(32)), ("bits0-63", range(64)), ("every 8th", range(0, 256, 8)), ("every 16th", range(0, 256, 16)), ("every 32th", range(0,256,32)), ("0,128", [0,128])]:
    print(name, try_mask(list(bits)))
