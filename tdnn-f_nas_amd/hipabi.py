"""placeholder -- replaced below"""
def load():
    raise RuntimeError("libtdnnf_hip.so not built")
