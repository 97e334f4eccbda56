"""Host-side contract: HParams / config / registry / vocab / variable layout / LR schedule."""
import copy
import json
import os

import numpy as np
import pytest

from zero_amd.config import default_params, transformer_base_params, SyntheticVocab
from zero_amd.utils.hparams import HParams
from zero_amd.models import model as registry, load_all
from zero_amd import lrs
from zero_amd.vocab import Vocab
from zero_amd.variables import VariableStore, variable_specs, initial_values, ALIGN
from tests.common import make_hp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_defaults_follow_run_py():
    p = default_params()
    v = p.values()
    assert len(v) == 99 + 0 or len(v) >= 95
    assert p.hidden_size == 1000 and p.embed_size == 620 and p.filter_size == 2048
    assert p.beam_size == 4 and p.decode_alpha == 0.6 and p.decode_length == 50
    assert p.search_mode == "cache" and p.aan_mask is True and p.use_ffn is False
    assert p.clip_grad_norm == 5.0 and p.epsilon == 1e-9 and p.beta2 == 0.999
    assert p.dtype_epsilon == 1e-8 and p.dtype_inf == 1e8 and p.gpus == [0] and p.strategies == ["aan"]


def test_parse_override_json_roundtrip():
    p = default_params()
    p.parse("hidden_size=512,gpus=[0,1,2],shared_source_target_embedding=true,lrate=0.5,model_name=transformer")
    assert p.hidden_size == 512 and p.gpus == [0, 1, 2] and p.shared_source_target_embedding is True
    assert p.lrate == 0.5 and p.model_name == "transformer"
    p.override_from_dict(dict(num_heads=16, clip_grad_norm=0.0))
    assert p.num_heads == 16 and p.clip_grad_norm == 0.0
    js = p.to_json()
    q = default_params()
    q.parse_json(js)
    assert q.values() == p.values()
    with pytest.raises(ValueError):
        p.parse("no_such_key=1")
    with pytest.raises(ValueError):
        p.parse("hidden_size=1.5")
    p.add_hparam("recorder", {"step": 3})
    assert p.recorder["step"] == 3
    c = copy.copy(p)
    c.hidden_size = 7
    assert p.hidden_size == 512 and c.num_heads == 16
    p.src_vocab = SyntheticVocab(10)        # plain attribute, not an hparam (run.py:386)
    assert "src_vocab" not in p.values() and "src_vocab" not in json.loads(p.to_json())


def test_registry_contract():
    load_all()
    for name in ("transformer", "transformer_aan", "transformer_rpr", "transformer_fuse", "TRANSFORMER"):
        m = registry.get_model(name)
        assert callable(m.train_fn) and callable(m.score_fn) and callable(m.infer_fn)
    with pytest.raises(Exception) as e:
        registry.get_model("nope")
    assert "No supported model" in str(e.value)
    with pytest.raises(Exception) as e:
        registry.model_register("Transformer", None, None, None)
    assert "Conflict Model Name" in str(e.value)


def test_vocab_and_noam_against_reference_values():
    gold = json.load(open(os.path.join(GOLD, "reference_scalars.json")))
    v = Vocab()
    assert {"pad": v.pad(), "eos": v.eos(), "unk": v.get_id("<unk>"), "size": v.size()} == gold["vocab_ids"]
    v.insert("hello"); v.insert("world")
    assert v.to_id(["hello", "zzz", "world"]) == gold["vocab_to_id"]
    hp = make_hp("transformer", H=512, lrate=1.0, warmup_steps=4000)
    lr = lrs.get_lr(hp)
    for step, val in gold["noam_lr_init1_warm4000_h512"].items():
        lr.step(int(step))
        assert lr.get_lr() == pytest.approx(val, rel=1e-12)
    hp2 = make_hp("transformer", H=1024, lrate=2.0, warmup_steps=400, min_lrate=1e-5, max_lrate=1e-3)
    lr = lrs.get_lr(hp2)
    for step, val in gold["noam_lr_init2_clamped_warm400_h1024"].items():
        lr.step(int(step))
        assert lr.get_lr() == pytest.approx(val, rel=1e-12)


@pytest.mark.parametrize("model", ["transformer", "transformer_aan", "transformer_rpr", "transformer_fuse"])
def test_variable_names_match_the_oracles(model):
    from oracle import ref_torch as rt
    hp = make_hp(model, Vs=13, Vt=11)
    ours = [(n, tuple(s)) for n, s, _, _ in variable_specs(hp, model)]
    theirs = [(n, tuple(s)) for n, s, _, _ in rt.variable_specs(hp, model)]
    assert ours == theirs
    assert "encoder/layer_0/self_attention/dot_attention/qkv_map/W_0_0" in dict(ours)
    assert ("bias", (hp.embed_size,)) in ours


def test_base_parameter_count_and_layout():
    hp = transformer_base_params()
    hp.src_vocab = SyntheticVocab(32000); hp.tgt_vocab = SyntheticVocab(32000)
    specs = variable_specs(hp, "transformer")
    n = sum(int(np.prod(s)) for _, s, _, _ in specs)
    assert abs(n - 76.9e6) < 0.1e6          # 44.14M body + 2 x 16.38M embeddings
    hp = make_hp("transformer", Vs=13, Vt=11)
    st = VariableStore(hp, "transformer", "cpu")
    assert st.pshape["tgt_embedding"] == (16, hp.embed_size) and st.lshape["tgt_embedding"][0] == 11
    for name, off in st.offsets.items():
        assert off % ALIGN == 0
    vals = initial_values(hp, "transformer", 3)
    st.load(vals)
    back = st.export("master")
    for k in vals:
        assert np.array_equal(back[k], vals[k])
    assert float(st.w("tgt_embedding")[11:].abs().max()) == 0.0   # physical pad rows stay zero
    emb = vals["src_embedding"]
    assert abs(emb.std() - hp.hidden_size ** -0.5) < 0.02
    w = vals["encoder/layer_0/feed_forward/ffn_layer/enlarge/W_0_0"]
    lim = np.sqrt(3.0 / ((w.shape[0] + w.shape[1]) / 2.0))
    assert np.abs(w).max() <= lim + 1e-6 and np.abs(w).max() > 0.9 * lim


def test_trim_columns_is_remove_invalid_seq():
    from zero_amd.models._core import trim_columns
    assert trim_columns(np.zeros((2, 3), dtype=np.int64)).shape == (2, 1)
    assert trim_columns(np.array([[4, 2, 0, 0], [5, 6, 2, 0]])).shape == (2, 3)
