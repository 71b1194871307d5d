"""Websocket policy server speaking the openpi client protocol (scripts/serve_policy.py:88-107 ->
openpi.serving.websocket_policy_server, absent submodule: [UPSTREAM-RECALL] SURVEY.md §8f rank 1).

Protocol: after the websocket handshake the server sends one binary frame with the msgpack-numpy packed policy metadata;
then, per request, the client sends a packed observation dict and receives the packed result of `policy.infer(obs)` plus
`server_timing` ({"infer_ms", "prev_total_ms"}).  An exception inside `infer` is reported as a text frame with the traceback
followed by a close frame with code 1011.  `GET /healthz` answers a plain `200 OK`.

The `websockets` package is not part of this image, so the RFC 6455 subset the protocol needs (handshake, masked client
frames, fragmentation, ping / pong / close, unmasked server frames) is implemented on asyncio streams directly.  Requests are
served one at a time per connection and inference itself is serialised by a lock: one GPU, batch-1 latency path.
"""
from __future__ import annotations

import asyncio
import base64
import hashlib
import logging
import struct
import time
import traceback

from lap_amd.policy_io import packb, unpackb

_GUID = "258EAFA5-E914-47DA-95CA-C5AB0DC85B11"
OP_CONT, OP_TEXT, OP_BINARY, OP_CLOSE, OP_PING, OP_PONG = 0x0, 0x1, 0x2, 0x8, 0x9, 0xA
INTERNAL_ERROR = 1011
log = logging.getLogger("lap_amd.serve_ws")


class ConnectionClosed(Exception):
    pass


def encode_frame(opcode: int, payload: bytes, mask: bytes | None = None) -> bytes:
    """One unfragmented frame; `mask` (4 bytes) only for client -> server frames."""
    n = len(payload)
    head = bytes([0x80 | opcode])
    mbit = 0x80 if mask else 0
    if n < 126:
        head += bytes([mbit | n])
    elif n < (1 << 16):
        head += bytes([mbit | 126]) + struct.pack("!H", n)
    else:
        head += bytes([mbit | 127]) + struct.pack("!Q", n)
    if mask:
        payload = bytes(b ^ mask[i & 3] for i, b in enumerate(payload)) if n < 4096 else _mask_fast(payload, mask)
        head += mask
    return head + payload


def _mask_fast(payload: bytes, mask: bytes) -> bytes:
    import numpy as np
    a = np.frombuffer(payload, dtype=np.uint8)
    m = np.frombuffer((mask * (len(payload) // 4 + 1))[:len(payload)], dtype=np.uint8)
    return (a ^ m).tobytes()


async def read_message(reader: asyncio.StreamReader, writer: asyncio.StreamWriter | None = None, *, expect_mask: bool = True):
    """-> (opcode, payload) of the next complete data message; control frames are answered / raised in between."""
    buf, first = b"", None
    while True:
        b0, b1 = await reader.readexactly(2)
        fin, opcode = b0 & 0x80, b0 & 0x0F
        masked, n = b1 & 0x80, b1 & 0x7F
        if n == 126:
            n, = struct.unpack("!H", await reader.readexactly(2))
        elif n == 127:
            n, = struct.unpack("!Q", await reader.readexactly(8))
        if expect_mask and not masked:
            raise ConnectionClosed("client frames must be masked")
        mask = await reader.readexactly(4) if masked else None
        data = await reader.readexactly(n)
        if mask:
            data = _mask_fast(data, mask)
        if opcode == OP_CLOSE:
            if writer is not None:
                writer.write(encode_frame(OP_CLOSE, data[:2]))
                await writer.drain()
            raise ConnectionClosed("close frame")
        if opcode == OP_PING:
            if writer is not None:
                writer.write(encode_frame(OP_PONG, data))
                await writer.drain()
            continue
        if opcode == OP_PONG:
            continue
        if opcode != OP_CONT:
            first = opcode
        buf += data
        if fin:
            return first, buf


class WebsocketPolicyServer:
    def __init__(self, policy, host: str = "0.0.0.0", port: int = 8000, metadata: dict | None = None):
        self._policy, self._host, self._port = policy, host, port
        self._metadata = metadata or {}
        self._infer_lock = asyncio.Lock()
        self._server: asyncio.AbstractServer | None = None

    @property
    def port(self) -> int:
        return self._server.sockets[0].getsockname()[1] if self._server is not None else self._port

    def serve_forever(self) -> None:
        asyncio.run(self.run())

    async def start(self):
        self._server = await asyncio.start_server(self._connection, self._host, self._port)
        return self

    async def run(self):
        await self.start()
        async with self._server:
            await self._server.serve_forever()

    async def close(self):
        if self._server is not None:
            self._server.close()
            await self._server.wait_closed()

    async def _handshake(self, reader, writer) -> bool:
        head = (await reader.readuntil(b"\r\n\r\n")).decode("latin1")
        lines = head.split("\r\n")
        path = lines[0].split(" ")[1] if len(lines[0].split(" ")) > 1 else "/"
        headers = {k.strip().lower(): v.strip() for k, v in (ln.split(":", 1) for ln in lines[1:] if ":" in ln)}
        if path == "/healthz":
            writer.write(b"HTTP/1.1 200 OK\r\nContent-Type: text/plain\r\nContent-Length: 3\r\nConnection: close\r\n\r\nOK\n")
            await writer.drain()
            return False
        key = headers.get("sec-websocket-key")
        if headers.get("upgrade", "").lower() != "websocket" or key is None:
            writer.write(b"HTTP/1.1 400 Bad Request\r\nConnection: close\r\nContent-Length: 0\r\n\r\n")
            await writer.drain()
            return False
        accept = base64.b64encode(hashlib.sha1((key + _GUID).encode()).digest()).decode()
        writer.write(("HTTP/1.1 101 Switching Protocols\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                      f"Sec-WebSocket-Accept: {accept}\r\n\r\n").encode())
        await writer.drain()
        return True

    async def _connection(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter):
        peer = writer.get_extra_info("peername")
        try:
            if not await self._handshake(reader, writer):
                return
            log.info("Connection from %s opened", peer)
            writer.write(encode_frame(OP_BINARY, packb(self._metadata)))
            await writer.drain()
            prev_total = None
            while True:
                start = time.monotonic()
                _, payload = await read_message(reader, writer)
                obs = unpackb(payload)
                try:
                    async with self._infer_lock:
                        t0 = time.monotonic()
                        action = await asyncio.get_running_loop().run_in_executor(None, self._policy.infer, obs)
                        infer_s = time.monotonic() - t0
                except Exception:
                    writer.write(encode_frame(OP_TEXT, traceback.format_exc().encode()))
                    writer.write(encode_frame(OP_CLOSE, struct.pack("!H", INTERNAL_ERROR)
                                              + b"Internal server error. Traceback included in previous frame."))
                    await writer.drain()
                    log.exception("policy.infer failed for %s; connection closed with code %d", peer, INTERNAL_ERROR)
                    return
                action = dict(action)
                action["server_timing"] = {"infer_ms": infer_s * 1000}
                if prev_total is not None:
                    action["server_timing"]["prev_total_ms"] = prev_total * 1000
                writer.write(encode_frame(OP_BINARY, packb(action)))
                await writer.drain()
                prev_total = time.monotonic() - start
        except (ConnectionClosed, asyncio.IncompleteReadError, ConnectionResetError):
            log.info("Connection from %s closed", peer)
        finally:
            writer.close()


def main(argv=None):
    """python -m lap_amd.serve_ws --config lap_libero --checkpoint-dir DIR --tokenizer-model paligemma_tokenizer.model
    (scripts/serve_policy.py: env / policy selection reduced to an explicit config + checkpoint; `--type ar` for LAP_AR)."""
    import argparse
    import dataclasses

    from lap_amd.config import get_config
    from lap_amd.serve import create_trained_policy, create_trained_policy_ar

    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="lap")
    ap.add_argument("--checkpoint-dir", required=True)
    ap.add_argument("--tokenizer-model", required=True, help="PaliGemma SentencePiece model file (not downloadable offline)")
    ap.add_argument("--type", choices=["flow", "ar"], default="flow")
    ap.add_argument("--default-prompt", default=None)
    ap.add_argument("--port", type=int, default=8000)
    a = ap.parse_args(argv)
    cfg = get_config(a.config)
    cfg = dataclasses.replace(cfg, model=dataclasses.replace(cfg.model, stop_action_to_vlm_grad=False))   # serve_policy.py:77-79
    make = create_trained_policy_ar if a.type == "ar" else create_trained_policy
    policy = make(cfg, a.checkpoint_dir, tokenizer_model_path=a.tokenizer_model, default_prompt=a.default_prompt)
    logging.basicConfig(level=logging.INFO)
    WebsocketPolicyServer(policy, "0.0.0.0", a.port, metadata=policy.metadata).serve_forever()


if __name__ == "__main__":
    main()
