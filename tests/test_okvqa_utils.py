"""OK-VQA answer post-processing (ref:icv_src/metrics/okvqa_utils.py:187-215, ref:utils.py:128-133) on the CPU.

nltk and inflection are not installed here, so the stemmer's RULE ORDER is pinned with stand-in modules in sys.modules: a
tokenizer that splits on blanks, a tagger that calls every word ending in "s" a plural noun, a lemmatizer and a singularizer that
mark what they touched.  What is pinned: the exception table wins over both rules (including its identity entries such as
"christmas"), "...ing" words go to the verb lemmatizer and never to the singularizer, plural nouns are singularized, everything
else passes through, and the generation is cut at the next prompt keyword / first ", " before stemming."""
import json
import sys
import types

import pytest


@pytest.fixture()
def stub_nlp(monkeypatch):
    nltk = types.ModuleType("nltk")
    nltk.tokenize = types.SimpleNamespace(word_tokenize=lambda text: text.split())
    nltk.pos_tag = lambda words: [(w, "NNS" if w.endswith("s") else ("VBG" if w.endswith("ing") else "NN")) for w in words]
    stem = types.ModuleType("nltk.stem")

    class WordNetLemmatizer:
        def lemmatize(self, word, pos):
            assert pos == "v"
            return f"lemma({word})"
    stem.WordNetLemmatizer = WordNetLemmatizer
    nltk.stem = stem
    inflection = types.ModuleType("inflection")
    inflection.singularize = lambda w: f"singular({w})"
    monkeypatch.setitem(sys.modules, "nltk", nltk)
    monkeypatch.setitem(sys.modules, "nltk.stem", stem)
    monkeypatch.setitem(sys.modules, "inflection", inflection)
    from icv_src.metrics import okvqa_utils as U
    monkeypatch.setattr(U, "stemmer", U.OKVQAStemmer())
    return U


def test_exception_table_ships_as_data_and_matches_the_published_mapping():
    from icv_src.metrics import okvqa_utils as U
    table = U.load_manual_matches()
    assert len(table) == 168
    # entries the automatic rules would get wrong: table words, identity entries, possessives
    for word, stem in {"weddings": "wed", "settings": "set", "minerals": "miner", "christmas": "christmas", "leaves": "leaf",
                       "riding": "ride", "police": "police", "hell's": "hell", "morning": "morn", "earing": "ear"}.items():
        assert table[word] == stem


def test_stemmer_rule_order(stub_nlp):
    U = stub_nlp
    s = U.stemmer.stem
    assert s("weddings") == "wed" and s("christmas") == "christmas"            # table first (plural rule would have fired)
    assert s("riding") == "ride" and s("morning") == "morn"                    # table before the "ing" rule
    assert s("running") == "lemma(running)"                                    # "ing" -> verb lemma ...
    assert s("rings") == "singular(rings)"                                     # plural noun -> singular
    assert s("kings swimming") == "singular(kings) lemma(swimming)"
    assert s("blue frisbee") == "blue frisbee"                                 # nothing applies
    assert s("weddings rings riding boats") == "wed singular(rings) ride singular(boats)"


def test_generation_is_cut_before_stemming(stub_nlp):
    U = stub_nlp
    assert U.postprocess_ok_vqa_generation("two dogs Question: what") == "two singular(dogs)"
    assert U.postprocess_ok_vqa_generation("surfing, skiing") == "lemma(surfing)"
    assert U.postprocess_ok_vqa_generation("weddings Short answer") == "wed"
    assert U.ok_vq_postprocess("frisbees\n Answer: x", "idefics-9b") == "singular(frisbees)"
    assert U.ok_vq_postprocess(" frisbees ", "open_flamingo") == "singular(frisbees)"
    assert U.ok_vq_postprocess("x", "other") is None


def test_missing_table_or_packages_raise_instead_of_changing_scores(monkeypatch, tmp_path, stub_nlp):
    U = stub_nlp
    monkeypatch.setenv("LICV_OKVQA_MANUAL_MATCHES", str(tmp_path / "absent.json"))
    with pytest.raises(FileNotFoundError, match="exception table"):
        U.OKVQAStemmer().stem("dogs")
    p = tmp_path / "t.json"
    p.write_text(json.dumps({"dogs": "hound"}))
    monkeypatch.setenv("LICV_OKVQA_MANUAL_MATCHES", str(p))
    assert U.OKVQAStemmer().stem("dogs cats") == "hound singular(cats)"
    monkeypatch.delenv("LICV_OKVQA_MANUAL_MATCHES")
    monkeypatch.setitem(sys.modules, "nltk", None)                              # import nltk -> ImportError
    with pytest.raises(ImportError, match="nltk"):
        U.OKVQAStemmer().stem("dogs")
