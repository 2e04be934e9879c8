"""nspeech_amd - MI355X-native Tacotron hot path (drop-in for neural_speech.models and
neural_speech.utils.audio of MLCogUP/nspeech).  HIP kernels live in csrc/ behind a C ABI."""
__version__ = "0.1.0"
