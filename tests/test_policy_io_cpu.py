"""Request-level policy transforms: prompt text / state bins against reference-generated fixtures,
normalisation arithmetic against hand-computed values, tokenizer masks with a tiny SentencePiece model trained on the spot,
the input / output stacks end to end, and the msgpack wire format."""
import json
import pathlib

import numpy as np
import pytest
import torch

from lap_amd import policy_io as pio
from lap_amd import prompt as P

GOLD = json.loads((pathlib.Path(__file__).parent / "golden" / "prompt_format_v1.json").read_text())


def test_prompt_text_matches_reference_fixture():
    reg = {"prompt": P.PROMPT_FORMAT_REGISTRY, "prediction": P.PREDICTION_PROMPT_FORMAT_REGISTRY, "vqa": {"default_vqa": P.DEFAULT_VQA_PROMPT_FORMAT}}
    assert len(GOLD["format"]) >= 100
    for c in GOLD["format"]:
        st = None if c["state_values"] is None else np.asarray(c["state_values"], dtype=np.float64)
        got = reg[c["registry"]][c["format"]].format_prompt(c["prompt"], st, c["state_type"], time_horizon_seconds=c["horizon"],
                                                            frame_description=c["frame"])
        assert got == c["expected"], (c["format"], c["prompt"], c["state"])
    lap = [c for c in GOLD["format"] if c["format"] == "lap" and c["state"] == "eef10"][0]
    assert lap["expected"].startswith("Task: ") and "; State: " in lap["expected"] and lap["expected"].endswith("; Answer: ")


def test_state_text_and_token_classes_match_reference_fixture():
    for c in GOLD["state"]:
        got = P.dataclasses.replace(P.STATE_TEXTS[c["template"]], min_dim=c["min_dim"]).render(np.asarray(c["state_values"], dtype=np.float64))
        assert got == c["expected"], c
    assert GOLD["pieces"] == P.CHECKER_PIECES
    for name, fn in P.CHECKERS.items():
        assert [bool(fn(p)) for p in P.CHECKER_PIECES] == GOLD["checkers"][name], name
    with pytest.raises(ValueError, match="Unknown prompt format"):
        P.resolve_prompt_format("nope")


def test_state_bins_edges():
    st = P.StateText(min_dim=0)
    idx = st.bin_indices(np.array([-1.0, -1.0 + 2 / 256, 0.0, 1.0 - 2 / 256, 1.0, 2.0, -2.0]))
    assert idx.tolist() == [0, 1, 128, 255, 255, 255, -1]


def test_normalize_unnormalize_arithmetic():
    stats = {"state": {"mean": [1.0, 2.0], "std": [0.5, 0.0], "q01": [0.0, 3.0], "q99": [2.0, 3.0], "min": [0.0, 5.0], "max": [4.0, 5.0]},
             "actions": {"mean": [0.0, 1.0], "std": [1.0, 2.0], "q01": [-1.0, 0.0], "q99": [1.0, 4.0], "min": [-2.0, 0.0], "max": [2.0, 8.0]}}
    x = {"state": np.array([1.5, 3.0]), "actions": np.array([[0.5, 2.0]]), "other": np.array([7.0])}
    z = pio.Normalize(stats, "normal")(x)
    np.testing.assert_allclose(z["state"], [(1.5 - 1) / (0.5 + 1e-6), (3 - 2) / 1e-6])
    assert z["other"][0] == 7.0
    with pytest.raises(Exception):
        pio.Normalize(stats, "normal")({"state": np.array([1.0, 2.0, 3.0])})      # data wider than the statistics: shape error, as upstream
    q = pio.Normalize(stats, "bounds_q99")({"state": np.array([3.0, 3.0]), "actions": np.array([[2.0, 2.0]])})
    np.testing.assert_allclose(q["state"], [(3.0 / (2 + 1e-6)) * 2 - 1, 0.0])      # not clipped; constant dimension -> 0
    np.testing.assert_allclose(q["actions"], [[(3.0 / (2 + 1e-6)) * 2 - 1, (2.0 / (4 + 1e-6)) * 2 - 1]])
    b = pio.Normalize(stats, "bounds")({"state": np.array([6.0, 5.0])})
    np.testing.assert_allclose(b["state"], [1.0, 0.0])                             # clipped; constant dimension -> 0
    # un-normalisation pads the statistics to the model's wider action vector
    a = np.array([[0.5, -1.0, 0.25, 0.75]])
    un = pio.Unnormalize(stats, "normal")({"actions": a})["actions"]
    np.testing.assert_allclose(un, [[0.5 * (1 + 1e-6), -1.0 * (2 + 1e-6) + 1.0, 0.25 * (1 + 1e-6), 0.75 * (1 + 1e-6)]])
    uq = pio.Unnormalize(stats, "bounds_q99")({"actions": a})["actions"]
    np.testing.assert_allclose(uq, [[(1.5 / 2) * (2 + 1e-6) - 1, 0.0, 0.25, 0.75]])
    ub = pio.Unnormalize(stats, "bounds")({"actions": a})["actions"]
    np.testing.assert_allclose(ub, [[(1.5 / 2) * (4 + 1e-8) - 2, 0.0, (1.25 / 2) * (2 + 1e-8) - 1, (1.75 / 2) * (2 + 1e-8) - 1]])
    # round trip on the recorded dimensions
    v = np.array([[0.3, 1.7]])
    for t in ("normal", "bounds_q99"):
        back = pio.Unnormalize(stats, t)({"actions": pio.Normalize(stats, t)({"actions": v})["actions"]})["actions"]
        np.testing.assert_allclose(back, v, atol=1e-9)
    with pytest.raises(ValueError, match="quantile stats"):
        pio.Normalize({"state": {"mean": [0.0], "std": [1.0]}}, "bounds_q99")
    with pytest.raises(ValueError, match="Unknown normalization type"):
        pio.Normalize(stats, "zscore")
    assert pio.Normalize(None)(x) is x


@pytest.fixture(scope="module")
def tiny_tokenizer():
    from tests.common import tiny_sentencepiece_proto
    return pio.PaligemmaTokenizer(model_proto=tiny_sentencepiece_proto(), max_len=48)


def test_tokenizer_masks(tiny_tokenizer):
    tk = tiny_tokenizer
    sp = tk._tokenizer
    state = np.array([0.1, -0.5, 0.9, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    toks, attn, reason, num, direc, loss = tk.tokenize("pick up the block", None, state)
    text = P.LAP_PROMPT_FORMAT.format_prompt("pick up the block", state)
    ids = sp.encode(text, add_bos=True)
    n = min(len(ids), 48)
    assert toks.dtype == np.int32 and toks.shape == (48,) and toks[0] == sp.bos_id() and toks[:n].tolist() == ids[:n]
    assert attn[:n].all() and not attn[n:].any() and (toks[n:] == sp.pad_id()).all()
    assert reason is None and num is None and direc is None and loss.all()
    # training: language action appended with EOS, masks over exactly those positions
    tk2 = pio.PaligemmaTokenizer(model_proto=sp.serialized_model_proto(), max_len=160)
    toks, attn, reason, num, direc, loss = tk2.tokenize("pick up the block", "move right 3 cm\nand open_gripper", state)
    n0 = len(sp.encode(text, add_bos=True))
    tail = sp.encode("move right 3 cm and open gripper", add_eos=True)
    assert toks[n0:n0 + len(tail)].tolist() == tail and toks[n0 + len(tail) - 1] == sp.eos_id()
    assert reason.sum() == len(tail) and reason[n0] and not reason[n0 - 1] and attn.sum() == n0 + len(tail)
    pieces = [sp.id_to_piece(int(t)) for t in toks]
    assert [bool(x) for x in num] == [bool(reason[i] and P.is_number(pieces[i])) for i in range(160)]
    assert [bool(x) for x in direc] == [bool(reason[i] and P.is_direction_natural(pieces[i])) for i in range(160)]
    assert num.any() and direc.any()
    # truncation keeps the masks inside max_len
    tk3 = pio.PaligemmaTokenizer(model_proto=sp.serialized_model_proto(), max_len=n0 + 2)
    toks, attn, reason, *_ = tk3.tokenize("pick up the block", "move right 3 cm", state)
    assert attn.all() and reason.sum() == 2 and toks.shape == (n0 + 2,)
    assert tk.decode(np.array([sp.bos_id(), *sp.encode("open gripper"), sp.eos_id(), 10**6, -1])) == "open gripper"
    with pytest.raises(ValueError, match="SentencePiece model"):
        pio.PaligemmaTokenizer()


def test_input_and_output_stacks(tiny_tokenizer):
    stats = {"state": {"mean": [0.0] * 8, "std": [1.0] * 8, "q01": [-2.0] * 8, "q99": [2.0] * 8},
             "actions": {"mean": [0.0] * 7, "std": [1.0] * 7, "q01": [-0.5] * 7, "q99": [0.5] * 7}}
    rs = np.random.RandomState(0)
    req = {"observation": {"base_0_rgb": rs.rand(3, 224, 224).astype(np.float32), "state": rs.uniform(-1, 1, 8)},
           "prompt": b"put_the cup on the plate.", "frame_description": "end-effector frame"}
    stack = pio.compose([pio.InjectDefaultPrompt("unused"), pio.CoTInputs(action_dim=32), pio.Normalize(stats, "bounds_q99"),
                         pio.TokenizePromptAndReasoning(tiny_tokenizer, discrete_state_input=True), pio.PadStatesAndActions(32)])
    out = stack(req)
    assert out["image"]["base_0_rgb"].shape == (224, 224, 3) and out["image"]["base_0_rgb"].dtype == np.uint8
    assert out["image_mask"]["base_0_rgb"] and not out["image_mask"]["left_wrist_0_rgb"] and not out["image"]["left_wrist_0_rgb"].any()
    np.testing.assert_allclose(out["state"][:8], (req["observation"]["state"] + 2) / (4 + 1e-6) * 2 - 1)
    assert out["state"].shape == (32,) and not out["state"][8:].any()
    expect = P.LAP_PROMPT_FORMAT.format_prompt("put_the cup on the plate.", out["state"][:8], frame_description="end-effector frame")
    assert "put the cup on the plate, predict the robot's action in the end-effector frame" in expect
    ids = tiny_tokenizer._tokenizer.encode(expect, add_bos=True)[:48]
    assert out["tokenized_prompt"][:len(ids)].tolist() == ids
    assert out["tokenized_langact_mask"] is None and out["tokenized_dataset_name"].shape == (100,) and out["sample_mask"] is True
    assert "prompt" not in out and "frame_description" not in out
    # default prompt is injected only when the request has none; r1_lite prompts keep the text after the last '@'
    o2 = pio.compose([pio.InjectDefaultPrompt("do something"), pio.CoTInputs(action_dim=32)])(
        {"observation": {"base_0_rgb": np.ones((224, 224, 3), np.uint8), "state": np.zeros(8)}})
    assert o2["prompt"] == "do something"
    o3 = pio.CoTInputs(action_dim=32)({"observation": {"base_0_rgb": np.ones((224, 224, 3), np.uint8), "left_wrist_0_rgb": np.ones((224, 224, 3), np.uint8),
                                                         "state": np.zeros(8)}, "prompt": "a@b@fold the towel", "dataset_name": b"r1_lite_x",
                                       "actions": np.ones((4, 7))})
    assert o3["prompt"] == "fold the towel" and o3["image_mask"]["left_wrist_0_rgb"] and o3["actions"].shape == (4, 32)
    v = pio.CoTInputs(action_dim=32)({"observation": {"state": np.zeros(8)}, "prompt": "x", "is_vqa_sample": True})   # VQA sample without caption
    assert v["language_actions"] == "" and v["sample_mask"] is True and v["is_vqa_sample"]
    # outputs: model-space action chunk [50, 32] -> client units on the 7 recorded dimensions, rest untouched
    acts = rs.uniform(-1, 1, (50, 32))
    res = pio.compose([pio.Unnormalize(stats, "bounds_q99"), pio.CoTOutputs()])({"state": out["state"], "actions": acts})
    np.testing.assert_allclose(res["actions"][:, :7], (acts[:, :7] + 1) / 2 * (1 + 1e-6) - 0.5)
    np.testing.assert_array_equal(res["actions"][:, 7:], acts[:, 7:])
    assert res["reasoning"] is None and set(res) == {"actions", "reasoning"}


def test_msgpack_numpy_round_trip():
    msg = {"actions": np.arange(12, dtype=np.float32).reshape(3, 4), "flag": np.bool_(True), "n": np.int64(7), "text": "hi",
           "nested": {"img": np.zeros((2, 2, 3), np.uint8)}}
    back = pio.unpackb(pio.packb(msg))
    np.testing.assert_array_equal(back["actions"], msg["actions"])
    assert back["actions"].dtype == np.float32 and back["nested"]["img"].shape == (2, 2, 3) and back["text"] == "hi"
    assert back["n"] == 7 and isinstance(back["n"], np.int64) and back["flag"] == np.bool_(True)
    with pytest.raises(ValueError, match="Unsupported dtype"):
        pio.packb({"x": np.array([1 + 2j])})


def test_resize_with_pad():
    from lap_amd.observation import resize_with_pad
    img = torch.full((2, 180, 320, 3), 0.5)
    out = resize_with_pad(img, 224, 224)
    assert out.shape == (2, 224, 224, 3)
    rh = int(180 / (320 / 224))           # 126 rows of picture, centred, -1 ("black") above and below
    top = (224 - rh) // 2
    assert torch.all(out[:, :top] == -1.0) and torch.all(out[:, top + rh:] == -1.0)
    assert torch.allclose(out[:, top:top + rh], torch.tensor(0.5), atol=1e-5)
    u8 = resize_with_pad(torch.full((1, 448, 224, 3), 200, dtype=torch.uint8), 224, 224)
    assert u8.dtype == torch.uint8 and torch.all(u8[:, :, :56] == 0) and torch.all(u8[:, :, 56:168] == 200) and torch.all(u8[:, :, 168:] == 0)
    same = torch.rand(1, 224, 224, 3)
    assert resize_with_pad(same, 224, 224) is same


def test_websocket_policy_server_round_trip():
    """The openpi client protocol against a stand-in policy: metadata frame first, packed observation in, packed result +
    server_timing out (masked / fragmented client frames, ping), traceback + close 1011 on failure, /healthz."""
    import asyncio
    import base64
    import hashlib
    import os

    from lap_amd import serve_ws as W

    class FakePolicy:
        metadata = {"robot": "test", "dims": np.arange(3)}

        def infer(self, obs):
            if "boom" in obs:
                raise RuntimeError("policy failed")
            return {"actions": obs["observation"]["state"][None].repeat(4, 0) * 2.0, "reasoning": None}

    async def client(port, messages, split=False):
        r, w = await asyncio.open_connection("127.0.0.1", port)
        key = base64.b64encode(os.urandom(16)).decode()
        w.write((f"GET / HTTP/1.1\r\nHost: 127.0.0.1:{port}\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                 f"Sec-WebSocket-Key: {key}\r\nSec-WebSocket-Version: 13\r\n\r\n").encode())
        head = (await r.readuntil(b"\r\n\r\n")).decode()
        assert head.startswith("HTTP/1.1 101") and base64.b64encode(hashlib.sha1((key + W._GUID).encode()).digest()).decode() in head
        got = [await W.read_message(r, expect_mask=False)]
        for m in messages:
            data = pio.packb(m)
            w.write(W.encode_frame(W.OP_PING, b"hi", mask=os.urandom(4)))
            if split:   # fragmented message: first half as BINARY without FIN, second half as CONTINUATION with FIN
                h = len(data) // 2
                f1 = bytearray(W.encode_frame(W.OP_BINARY, data[:h], mask=os.urandom(4))); f1[0] &= 0x7F
                f2 = bytearray(W.encode_frame(W.OP_CONT, data[h:], mask=os.urandom(4)))
                w.write(bytes(f1) + bytes(f2))
            else:
                w.write(W.encode_frame(W.OP_BINARY, data, mask=os.urandom(4)))
            await w.drain()
            got.append(await W.read_message(r, expect_mask=False))
            if got[-1][0] == W.OP_TEXT:   # error report: the close frame follows
                with pytest.raises(W.ConnectionClosed):
                    await W.read_message(r, expect_mask=False)
                got.append("closed")
                break
        w.close()
        return got

    async def scenario():
        srv = await W.WebsocketPolicyServer(FakePolicy(), "127.0.0.1", 0, metadata=FakePolicy.metadata).start()
        port = srv.port
        obs = {"observation": {"state": np.linspace(-1, 1, 7), "base_0_rgb": np.zeros((8, 8, 3), np.uint8)}, "prompt": "x"}
        big = {"observation": {"state": np.ones(7), "base_0_rgb": np.random.RandomState(0).randint(0, 255, (224, 224, 3)).astype(np.uint8)}}
        a = await client(port, [obs, obs, big])
        b = await client(port, [obs], split=True)
        c = await client(port, [{"boom": 1}, obs])
        r, w = await asyncio.open_connection("127.0.0.1", port)
        w.write(b"GET /healthz HTTP/1.1\r\nHost: x\r\n\r\n"); await w.drain()
        health = await r.read()
        w.close()
        await srv.close()
        return a, b, c, health

    a, b, c, health = asyncio.run(scenario())
    meta = pio.unpackb(a[0][1])
    assert a[0][0] == W.OP_BINARY and meta["robot"] == "test" and meta["dims"].tolist() == [0, 1, 2]
    r1, r2, r3 = (pio.unpackb(m[1]) for m in a[1:])
    np.testing.assert_allclose(r1["actions"], np.linspace(-1, 1, 7)[None].repeat(4, 0) * 2)
    assert "infer_ms" in r1["server_timing"] and "prev_total_ms" not in r1["server_timing"] and "prev_total_ms" in r2["server_timing"]
    assert r3["actions"].shape == (4, 7) and r1["reasoning"] is None
    np.testing.assert_allclose(pio.unpackb(b[1][1])["actions"], r1["actions"])
    assert c[1][0] == W.OP_TEXT and b"RuntimeError: policy failed" in c[1][1] and c[2] == "closed"
    assert health.startswith(b"HTTP/1.1 200 OK") and health.endswith(b"OK\n")


def test_language_actions_match_reference_fixture():
    """Numbers <-> text and the frame transforms against outputs of the reference's own modules (tests/golden/make_lang_action_golden.py)."""
    from lap_amd import lang_actions as LA
    G = json.loads((pathlib.Path(__file__).parent / "golden" / "lang_action_v1.json").read_text())
    T = LA.case_tables()
    texts = G["texts"]
    assert texts[:len(T["texts"])] == T["texts"]
    for c in G["summaries"]:
        assert LA.summarize_numeric_actions(T["chunks"][c["chunk"]], c["sum_decimal"], c["rot"]) == c["expected"], c
    for c in G["bimanual"]:
        ch = np.asarray(T["chunks"][c["chunk"]])
        two = np.concatenate([ch, ch[::-1] * 0.5], axis=1)
        assert LA.summarize_bimanual_numeric_actions(two, c["sum_decimal"], True) == c["expected"], c
    for c in G["scale"]:
        assert LA.describe_language_action_scale(texts[c["text"]]) == c["expected"], c
    for c in G["idle"]:
        assert LA.is_idle_language_action(texts[c["text"]], c["sum_decimal"], c["rot"]) == c["expected"], (texts[c["text"]], c)
    fmts = {"verbose_with_rotation": LA.VERBOSE_WITH_ROTATION_FORMAT, "verbose_eef_with_rotation": LA.VERBOSE_EEF_WITH_ROTATION_FORMAT,
            "compact_rot": LA.LanguageActionFormat(name="c", style="compact", include_rotation=True)}
    for c in G["parse"]:
        st = None if c["state"] is None else np.asarray(T["states"][c["state"]])
        mv, g = fmts[c["format"]].parse_language_to_deltas(texts[c["text"]], initial_state=st)
        np.testing.assert_allclose(mv, c["movement"], rtol=0, atol=1e-12, err_msg=str(c))
        assert g == c["gripper"], c
    for c in G["vla0"]:
        mv, g = LA.VLA0_CHUNKED_FORMAT.parse_language_to_deltas(texts[c["text"]])
        np.testing.assert_allclose(mv, c["movement"], atol=1e-15)
        assert g == c["gripper"]
        np.testing.assert_allclose(LA.VLA0_CHUNKED_FORMAT.parse_to_full_actions(texts[c["text"]]), c["full"], atol=1e-15)
    acts = np.asarray(T["chunks"][2])[:, :7]
    assert LA.VLA0_CHUNKED_FORMAT.summarize_actions(acts * 20) == G["vla0_summary"]["expected"]
    assert LA.VLA0ActionFormat().summarize_actions(acts[0] * 20) == G["vla0_summary"]["single"]
    for c in G["frames"]:
        a, st = np.asarray(T["frame_actions"][c["action"]]), np.asarray(T["states"][c["state"]])
        np.testing.assert_allclose(LA.transform_actions_from_eef_frame(a, st, c["dataset"]), c["from_eef"], atol=1e-12, err_msg=str(c))
        for wrist in (0, 1):
            if f"to_eef_{wrist}" in c:
                np.testing.assert_allclose(LA.transform_actions_to_eef_frame(a, st, c["dataset"], bool(wrist)), c[f"to_eef_{wrist}"], atol=1e-12)
    np.testing.assert_allclose(LA.transform_actions_from_eef_frame(np.asarray(T["frame_actions"]), np.asarray([T["states"][0]])), G["from_eef_chunk"], atol=1e-12)
    np.testing.assert_allclose(LA.rot6d_to_rotmat(np.asarray(T["states"][1])[3:9]), G["rot6d"], atol=1e-14)
    assert LA.get_language_action_format("vla0_chunked").get_sum_decimal() == "vla0" and LA.VERBOSE_WITH_ROTATION_FORMAT.get_sum_decimal() == "0f"
    with pytest.raises(ValueError, match="Unknown language action format"):
        LA.get_language_action_format("nope")


def test_cot_outputs_text_to_actions():
    from lap_amd import lang_actions as LA
    text = "move forward 3 cm, move left 2 cm, tilt left 10 degrees, open gripper"
    out = pio.CoTOutputs(language_action_format="verbose_with_rotation")({"reasoning": text, "tokens": np.zeros(3)})
    np.testing.assert_allclose(out["actions"], [0.03, 0.02, 0.0, np.deg2rad(10), 0.0, 0.0, 1.0])
    assert out["reasoning"] == text
    # end-effector-frame format: rotated into the base frame with the request's raw state ([xyz, euler, gripper])
    state = np.array([0.1, 0.2, 0.3, 0.0, 0.0, np.pi / 2, 0.5])     # yaw 90 degrees
    eef = pio.CoTOutputs(language_action_format="verbose_eef_with_rotation")({"reasoning": "move forward 10 cm", "raw_state": state})
    np.testing.assert_allclose(eef["actions"][:3], [0.0, 0.1, 0.0], atol=1e-12)      # eef x is base y after the yaw
    assert eef["actions"].shape == (6,)                                                # no gripper phrase -> no 7th value
    v = pio.CoTOutputs(language_action_format="vla0_chunked", transform_strategy="vla0", normalization_type="bounds_q99",
                       norm_stats={"actions": {"mean": [0.0] * 7, "std": [1.0] * 7, "q01": [-1.0] * 7, "q99": [3.0] * 7}})
    full = v({"reasoning": "500 0 1000 " * 30})["actions"]
    assert full.shape == (10, 7)
    np.testing.assert_allclose(full[0, :3], [(0.0 + 1) / 2 * (4 + 1e-6) - 1, -1.0, 3.0 + 1e-6], atol=1e-9)   # bins 500, 0, 1000 -> 0, -1, +1 -> [q01, q99]
    with pytest.raises(AssertionError):
        pio.CoTOutputs()({"reasoning": "move up 1 cm"})


def test_action_processor_and_training_inputs(tiny_tokenizer):
    from lap_amd import lang_actions as LA
    G = json.loads((pathlib.Path(__file__).parent / "golden" / "lang_action_v1.json").read_text())
    cases = LA.processor_cases()
    assert len(cases) == len(G["processor"])
    for c, g in zip(cases, G["processor"]):
        proc = LA.ActionProcessor(language_action_format=LA.get_language_action_format(c["format"]))
        text, frame = proc.summarize_language_actions({"language_actions": np.asarray(c["actions"]), **c["flags"]}, "language_actions",
                                                      None if c["state"] is None else np.asarray(c["state"]), c["dataset"], c["rotation_applied"])
        assert (text, frame) == (g["text"], g["frame"]), c
        m = LA.ActionProcessor.extract_motion_components(np.asarray(c["actions"]))
        assert {k: float(v) for k, v in m.items()} == pytest.approx(g["motion"], abs=1e-12)
    # training sample through the input stack: label text from raw deltas, frame description of the frame used, EOS-terminated
    c = cases[0]
    sample = {"observation": {"base_0_rgb": np.ones((224, 224, 3), np.uint8), "left_wrist_0_rgb": np.ones((224, 224, 3), np.uint8),
                              "state": np.zeros(8)}, "prompt": "stack the cups", "dataset_name": c["dataset"], "raw_state": np.asarray(c["state"]),
              "language_actions": np.asarray(c["actions"]), "actions": np.zeros((10, 7)), "has_wrist_image": True}
    tin = pio.CoTInputs(action_dim=32, language_action_format="verbose_eef_with_rotation")(sample)
    proc = LA.ActionProcessor(language_action_format=LA.VERBOSE_EEF_WITH_ROTATION_FORMAT)
    want, frame = proc.summarize_language_actions(sample, "language_actions", np.asarray(c["state"]), c["dataset"], False)
    assert tin["language_actions"] == want and tin["frame_description"] == frame == "end-effector frame" and tin["sample_mask"] is True
    tk = pio.PaligemmaTokenizer(model_proto=tiny_tokenizer._tokenizer.serialized_model_proto(), max_len=200)
    out = pio.TokenizePromptAndReasoning(tk, discrete_state_input=True)(tin)
    assert out["tokenized_langact_mask"].sum() == len(tk.encode(want, add_eos=True)) and "language_actions" not in out
    idle = dict(sample, language_actions=np.array([0.001, 0.0, 0.0, 0.01, 0.0, 0.0, 1.0]))
    assert pio.CoTInputs(action_dim=32)(idle)["sample_mask"] is False                       # < 1 cm and < 10 degrees
    rough = pio.CoTInputs(action_dim=32, use_rough_scale=True)(sample)
    assert rough["sample_mask"] is True and not any(ch.isdigit() for ch in rough["language_actions"])
    base = pio.CoTInputs(action_dim=32, language_action_format="verbose_with_rotation")(sample)
    assert base["frame_description"] == "robot base frame"
    off = pio.CoTInputs(action_dim=32, enable_langact_training=False)(sample)
    assert "language_actions" not in off and off["sample_mask"] is True
    np.random.seed(0)
    dropped = pio.CoTInputs(action_dim=32, wrist_image_dropout_prob=1.0)(sample)
    assert not dropped["image"]["left_wrist_0_rgb"].any() and not dropped["image_mask"]["left_wrist_0_rgb"]
    assert pio.CoTInputs(action_dim=32, wrist_image_dropout_prob=1.0, random_mask_prob=1.0)(sample)["image_mask"]["left_wrist_0_rgb"]


def test_normalisation_transforms_match_reference_generated_fixture():
    """tests/golden/normalize_v1.json (make_normalize_golden.py: the reference's `Normalize`, `Unnormalize` and the training side's
    `NormalizeActionAndProprio`, compiled from its own source and run on a seeded grid with a constant dimension, rows far outside the
    quantiles, 32-wide model outputs, a trajectory without state and the singular "action" group name)."""
    import json
    import pathlib

    fx = json.loads((pathlib.Path(__file__).parent / "golden" / "normalize_v1.json").read_text())
    stats = fx["stats"]
    xa, xs, wide = np.asarray(fx["x_actions"]), np.asarray(fx["x_state"]), np.asarray(fx["wide_actions"])
    for case in fx["cases"]:
        t = case["type"]
        n = pio.Normalize(stats, t)({"actions": xa.copy(), "state": xs.copy(), "other": np.ones(2)})
        for k, ref in case["normalize"].items():
            np.testing.assert_allclose(n[k], np.asarray(ref), rtol=1e-12, atol=1e-12, err_msg=f"Normalize {t} {k}")
        u = pio.Unnormalize(stats, t)({"actions": wide.copy(), "state": xs.copy()})
        for k, ref in case["unnormalize"].items():
            np.testing.assert_allclose(u[k], np.asarray(ref), rtol=1e-12, atol=1e-12, err_msg=f"Unnormalize {t} {k}")
        tr = pio.NormalizeActionAndProprio(stats, t)({"actions": xa.copy(), "state": xs.copy()})
        assert str(tr["actions"].dtype) == case["traj"]["dtype"] == "float32"
        np.testing.assert_array_equal(tr["actions"], np.asarray(case["traj"]["actions"], dtype=np.float32), err_msg=f"train-side {t} actions")
        np.testing.assert_array_equal(tr["state"], np.asarray(case["traj"]["state"], dtype=np.float32), err_msg=f"train-side {t} state")
        tr2 = pio.NormalizeActionAndProprio({"action": stats["actions"]}, t)({"actions": xa.copy()})
        np.testing.assert_array_equal(tr2["actions"], np.asarray(case["traj_no_state"]["actions"], dtype=np.float32))
    q = next(c for c in fx["cases"] if c["type"] == "bounds_q99")
    # the two sides differ exactly where it matters: beyond the quantiles the policy side extrapolates, the training side clips
    assert np.abs(np.asarray(q["normalize"]["actions"])).max() > 5.0 and np.abs(np.asarray(q["traj"]["actions"])).max() == 1.0


def test_tokenizer_and_tokenize_transform_match_reference_generated_fixture():
    """tests/golden/tokenize_v1.json (make_tokenize_golden.py): the reference's `PaligemmaTokenizer.tokenize` and its
    `TokenizePromptAndReasoning`, run unmodified on the tiny SentencePiece model the fixture carries — ids, all five masks, truncation,
    VQA / prediction formats, frame description, seeded reasoning dropout (np.random) and state dropout, the left-padded dataset name."""
    import base64
    import json
    import pathlib
    import random
    import zlib

    fx = json.loads((pathlib.Path(__file__).parent / "golden" / "tokenize_v1.json").read_text())
    proto = zlib.decompress(base64.b64decode(fx["sentencepiece_model_zlib_b64"]))
    ib = lambda a: None if a is None else np.asarray(a, dtype=bool)
    for c in fx["tokenize"]:
        a = dict(c["args"])
        tk = pio.PaligemmaTokenizer(model_proto=proto, max_len=a.pop("max_len"), reasoning_mask_prob=a.pop("p"))
        seed = a.pop("seed")
        if seed is not None:
            np.random.seed(seed); random.seed(seed)
        st = None if a["state"] is None else np.asarray(a.pop("state"))
        a.pop("state", None)
        toks, attn, reason, num, direc, loss = tk.tokenize(a.pop("prompt"), a.pop("reasoning"), st, **a)
        assert toks.dtype == np.int32 and toks.tolist() == c["tokens"], c["args"]
        for got, key in ((attn, "attn"), (reason, "reason"), (num, "number"), (direc, "direction"), (loss, "loss")):
            if c[key] is None:
                assert got is None, (key, c["args"])
            else:
                np.testing.assert_array_equal(np.asarray(got, dtype=bool), ib(c[key]), err_msg=f"{key} {c['args']}")
    for c in fx["transform"]:
        sample = {k: (np.asarray(v) if isinstance(v, list) else v) for k, v in c["sample"].items()}
        tf = pio.TokenizePromptAndReasoning(pio.PaligemmaTokenizer(model_proto=proto, max_len=96), discrete_state_input=True, dataset_name_pad_len=20,
                                            verbose_mode=c["verbose"])
        out = tf(sample)
        assert sorted(out) == c["out_keys"]
        for k, ref in c["out"].items():
            got = out[k]
            if isinstance(ref, list):
                np.testing.assert_array_equal(np.asarray(got).astype(int), np.asarray(ref), err_msg=k)
            else:
                assert (got is None and ref is None) or got == ref, (k, got, ref)


def test_cot_inputs_match_reference_generated_fixture():
    """tests/golden/cot_inputs_v1.json (make_cot_inputs_golden.py): the reference's `CoTInputs`, imported unmodified, end to end — an
    inference request (CHW float image, byte prompt, frame description), training samples (label text in the end-effector / base frame, idle
    -> sample_mask False, rough scale, random base frame through python's `random`, language-action training off, wrist dropout + random
    un-masking through np.random), a VQA sample and a prediction sample.  Every key and every value of the output must agree."""
    import json
    import pathlib
    import random

    def load(v):
        if isinstance(v, dict):
            if "__nd__" in v:
                return np.asarray(v["__nd__"], dtype=v["dtype"])
            if "__bytes__" in v:
                return v["__bytes__"].encode()
            return {k: load(x) for k, x in v.items()}
        return v

    def same(got, ref, path):
        if isinstance(ref, dict):
            assert isinstance(got, dict) and sorted(got) == sorted(ref), (path, sorted(got) if isinstance(got, dict) else got, sorted(ref))
            for k in ref:
                same(got[k], ref[k], f"{path}.{k}")
        elif isinstance(ref, np.ndarray):
            g = np.asarray(got)
            assert g.dtype == ref.dtype and g.shape == ref.shape, (path, g.dtype, ref.dtype, g.shape, ref.shape)
            np.testing.assert_array_equal(g, ref, err_msg=path)
        elif isinstance(ref, (bool, type(None))):
            assert (got is None) == (ref is None) and bool(got) == bool(ref), (path, got, ref)
        elif isinstance(ref, float):
            assert float(got) == ref, (path, got, ref)
        else:
            assert got == ref, (path, got, ref)

    cases = json.loads((pathlib.Path(__file__).parent / "golden" / "cot_inputs_v1.json").read_text())
    assert len(cases) >= 10 and not any("error" in c for c in cases)
    for c in cases:
        np.random.seed(5); random.seed(5)
        out = pio.CoTInputs(**c["config"])(load(c["data"]))
        same(out, load(c["out"]), c["title"])


def test_cot_outputs_match_reference_generated_fixture():
    """tests/golden/cot_outputs_v1.json (make_cot_outputs_golden.py): the reference's `CoTOutputs`, imported unmodified — pass-through,
    text -> deltas in base / end-effector frame with and without the request's raw state, texts without a gripper command or without
    anything parseable, and the VLA-0 grid with its three un-normalisations (and an unknown type: untouched)."""
    import json
    import pathlib
    import types

    fx = json.loads((pathlib.Path(__file__).parent / "golden" / "cot_outputs_v1.json").read_text())
    state = np.asarray(fx["state"])
    st = types.SimpleNamespace(**{k: np.asarray(v) for k, v in fx["stats"].items()})
    n = {"flow": 0, "text": 0, "vla0": 0}
    for c in fx["cases"]:
        n[c["kind"]] += 1
        if c["kind"] == "flow":
            r = pio.CoTOutputs("verbose_eef_with_rotation")({"actions": np.asarray(c["actions"])})
            assert r["reasoning"] is None
            np.testing.assert_array_equal(r["actions"], np.asarray(c["out"]["actions"]))
            continue
        if c["kind"] == "text":
            data = {"actions": np.zeros((1, 7)), "reasoning": fx["texts"][c["text"]], **({"raw_state": state} if c["with_state"] else {})}
            r = pio.CoTOutputs(c["format"])(data)
        else:
            from lap_amd import lang_actions as la
            kw = {} if c["ntype"] is None else {"norm_stats": {"actions": st}, "normalization_type": c["ntype"]}
            r = pio.CoTOutputs(la.VLA0_CHUNKED_FORMAT, transform_strategy="vla0", **kw)({"actions": np.zeros((1, 7)), "reasoning": c["text"]})
        assert r["reasoning"] == c["out"]["reasoning"]
        np.testing.assert_allclose(np.asarray(r["actions"], dtype=np.float64), np.asarray(c["out"]["actions"]), rtol=1e-12, atol=1e-12, err_msg=str(c)[:120])
    assert n["flow"] == 1 and n["text"] >= 20 and n["vla0"] == 5
