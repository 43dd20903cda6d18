"""MI355X-native MOBODY hot path (ensemble-dynamics rollout + Q-weighted-BC policy update).

Import as `mobody_amd`.  Sub-modules:
  _lib      ctypes binding of csrc/libmobody_hip.so (the C ABI in include/mobody_hip.h)
  packing   reference state_dict tensors <-> packed weight blobs
  algo.*    host-side mirror of the reference's plugin interface (call_algo, MOBODY,
            MOBODYEnsembleDynamics, ReplayBuffer, termination functions)
There is no CPU fallback: importing `_lib` without the built HIP library raises.
"""
__version__ = "0.1.0"
