"""Generates tests/golden/text_golden.json by IMPORTING the reference's text front end.

Run once, in the build container only (the reference never travels to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_text_golden.py
python3.9 there has the real `unidecode` (1.2.0; the reference pins 1.0.22); `inflect` is not
installable, so it is stubbed and NO sentence below contains a digit (the digit branch of the
reference is therefore unpinned - see tests/test_text.py)."""
import json
import os
import sys
import types

stub = types.ModuleType("inflect")


class _Engine(object):
    def number_to_words(self, *a, **k):
        raise RuntimeError("digit path is not pinned by this fixture")


stub.engine = _Engine
sys.modules["inflect"] = stub
sys.path.insert(0, "/root/reference")
from neural_speech.utils.text import sequence_to_text, text_to_sequence  # noqa: E402
from neural_speech.utils.text.symbols import symbols  # noqa: E402

SENTENCES = [
    "Hello, World.",
    "Turn left on {HH AW1 S S T AH0 N} Street.",
    "Dr. Smith   met Mr. Jones!",
    "Mrs. Robinson and Drs. Who; Lt. Dan, Sgt. Pepper & Capt. Hook?",
    "The St. Louis Co. Ltd. was founded by Col. Sanders, Esq.",
    "Scientists at the CERN laboratory say they have discovered a new particle.",
    "There’s a way to measure the acute emotional intelligence that has never gone out of style.",
    "President Trump met with other leaders at the Group of Twenty conference.",
    "The Senate's bill to repeal and replace the Affordable Care-Act is now imperiled.",
    "Generative adversarial network or variational auto-encoder.",
    "The buses aren't the problem, they actually provide a solution.",
    "café naïve façade — “quoted” text… with dashes – and São Paulo",
    "UPPER lower MiXeD case\twith\ttabs\nand newlines",
    "{AY1 M} a {R OW1 B AA0 T} speaking {IH0 N} phonemes {ZZ} end",
    "Weird symbols: # % ^ * _ ~ [brackets] (parens) 'single' \"double\"",
    "",
    "   leading and trailing spaces   ",
    "Gen. Maj. Rev. Hon. Jr. Ft. Worth",
]

out = {"symbols": symbols, "cases": []}
for cleaners in (["english_cleaners"], ["basic_cleaners"], ["transliteration_cleaners"]):
    for s in SENTENCES:
        ids = text_to_sequence(s, cleaners)
        out["cases"].append({"text": s, "cleaners": cleaners, "ids": ids, "roundtrip": sequence_to_text(ids)})

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_golden.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0, ensure_ascii=True)
print("wrote", path, len(out["cases"]), "cases")
