"""TEST INFRASTRUCTURE: oracle camera restatement vs tests/golden/camera.json (reference-built)."""
import ctypes, json, struct, os, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'liboracle.so'))
g = json.load(open(os.path.join(here, '..', 'tests', 'golden', 'camera.json')))
class Cam(ctypes.Structure):
    _fields_=[('position',ctypes.c_float*3),('front',ctypes.c_float*3),('up',ctypes.c_float*3),('right',ctypes.c_float*3),('world_up',ctypes.c_float*3),('yaw',ctypes.c_float),('pitch',ctypes.c_float)]
def f(u): return struct.unpack('f',struct.pack('I',u))[0]
def u32(arr): return list(np.frombuffer(np.array(list(arr),dtype=np.float32).tobytes(),dtype=np.uint32))
bad=0
for c in g['cases']:
    cam=Cam(); pos=(ctypes.c_float*3)(*[f(x) for x in c['pos']])
    L.o_camera_init(ctypes.byref(cam),pos,ctypes.c_float(f(c['yaw'])),ctypes.c_float(f(c['pitch'])))
    ip=(ctypes.c_float*16)(); iv=(ctypes.c_float*16)(); cp=(ctypes.c_float*4)()
    L.o_camera_ubo(ctypes.byref(cam),c['width'],c['height'],ip,iv,cp)
    ok = u32(ip)==c['inv_proj'] and u32(iv)==c['inv_view'] and u32(cam.front)==c['front'] and u32(cam.right)==c['right'] and u32(cam.up)==c['up']
    bad += (not ok)
print('camera cases', len(g['cases']), 'bad', bad)
