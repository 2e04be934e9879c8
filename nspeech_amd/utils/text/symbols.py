"""Symbol inventory: pad, eos, 52 letters, 11 punctuation marks + space, 84 ARPAbet phones
prefixed with '@' (149 ids, same order as neural_speech/utils/text/symbols.py:9-17)."""
from . import cmudict

_pad = "_"
_eos = "~"
_characters = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz!'(),-.:;? "
_arpabet = ["@" + s for s in cmudict.valid_symbols]

symbols = [_pad, _eos] + list(_characters) + _arpabet
