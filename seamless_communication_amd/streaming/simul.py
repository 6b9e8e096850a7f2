"""The small part of SimulEval that the reference's streaming agents are written against.

The reference builds its streaming pipeline on ``simuleval`` (``GenericAgent`` / ``AgentPipeline`` / segments /
actions: imports in src/seamless_communication/streaming/agents/*.py).  simuleval is not installed here and not part
of /root/reference, so the contract those agents rely on is restated (simuleval 1.1 semantics, parity unpinned; the
agents written on top of it ARE pinned: the reference's agent classes run on this very module in
tests/golden/make_streaming_goldens.py and their traces are replayed on streaming/agents.py):

* a pipeline is a chain of agents; ``pushpop(segment)`` feeds one source segment to the first agent and hands every
  agent's output segment to the next one; the last agent's output is returned;
* ``push`` updates the agent's states from the segment (``states.update_source``), ``pop`` asks ``policy(states)`` for
  an action: a read gives an ``EmptySegment``; a write gives its content (wrapped into the agent's target segment type
  unless it already is a segment);
* speech segments extend ``states.source`` (samples / frames), text segments append to it.
"""
from __future__ import annotations

from argparse import Namespace
from dataclasses import dataclass, field
from typing import Any, List, Optional


# --------------------------------------------------------------------------- segments
@dataclass
class Segment:
    index: int = 0
    content: Any = None
    finished: bool = False
    is_empty: bool = False
    data_type: Optional[str] = None
    tgt_lang: Optional[str] = None
    config: dict = field(default_factory=dict)


@dataclass
class EmptySegment(Segment):
    is_empty: bool = True


@dataclass
class TextSegment(Segment):
    content: Any = ""
    data_type: Optional[str] = "text"


@dataclass
class SpeechSegment(Segment):
    content: Any = field(default_factory=list)
    sample_rate: int = -1
    data_type: Optional[str] = "speech"


SEGMENT_TYPE = {"text": TextSegment, "speech": SpeechSegment}


# --------------------------------------------------------------------------- actions
class Action:
    def is_read(self) -> bool:
        raise NotImplementedError


class ReadAction(Action):
    def is_read(self) -> bool:
        return True

    def __repr__(self) -> str:
        return "ReadAction()"


class WriteAction(Action):
    def __init__(self, content: Any, finished: bool) -> None:
        self.content = content
        self.finished = finished

    def is_read(self) -> bool:
        return False

    def __repr__(self) -> str:
        return f"WriteAction(finished={self.finished})"


# --------------------------------------------------------------------------- states
class AgentStates:
    """simuleval.agents.states.AgentStates with the reference's override that never touches ``target``
    (streaming/agents/common.py:25-28)."""

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self.source: Any = []
        self.target: List[Any] = []
        self.source_finished = False
        self.target_finished = False
        self.source_sample_rate = 0
        self.tgt_lang: Optional[str] = None

    def update_source(self, segment: Segment) -> None:
        self.source_finished = segment.finished
        if self.tgt_lang is None and segment.tgt_lang is not None:
            self.tgt_lang = segment.tgt_lang
        if not segment.is_empty:
            if isinstance(segment, SpeechSegment):
                self.source += list(segment.content) if not isinstance(segment.content, list) else segment.content
                self.source_sample_rate = segment.sample_rate
            else:
                self.source.append(segment.content)

    def update_target(self, segment: Segment) -> None:
        self.target_finished = segment.finished


# --------------------------------------------------------------------------- agents
class GenericAgent:
    source_type: Optional[str] = None
    target_type: Optional[str] = None

    def __init__(self, args: Optional[Namespace] = None) -> None:
        self.args = args if args is not None else Namespace()
        self.states = self.build_states()
        self.reset()

    def build_states(self) -> AgentStates:
        return AgentStates()

    def reset(self) -> None:
        self.states.reset()

    def policy(self, states: AgentStates) -> Action:
        raise NotImplementedError

    def push(self, source_segment: Segment, states: Optional[AgentStates] = None) -> None:
        (states if states is not None else self.states).update_source(source_segment)

    def pop(self, states: Optional[AgentStates] = None) -> Segment:
        states = states if states is not None else self.states
        if states.target_finished:
            return EmptySegment(finished=True)
        action = self.policy(states)
        if not isinstance(action, Action):
            raise RuntimeError(f"The return value of {type(self).__name__}.policy must be an Action, got {type(action)}")
        if action.is_read():
            return EmptySegment()
        if isinstance(action.content, Segment):
            return action.content
        segment = SEGMENT_TYPE[self.target_type or "text"](index=0, content=action.content, finished=action.finished)
        states.update_target(segment)
        return segment

    def pushpop(self, segment: Segment, states: Optional[AgentStates] = None) -> Segment:
        self.push(segment, states)
        return self.pop(states)


class AgentPipeline(GenericAgent):
    """A chain of agents behaving as one (simuleval.agents.AgentPipeline)."""

    def __init__(self, module_list: List[GenericAgent]) -> None:
        self.module_list = module_list
        self.source_type = module_list[0].source_type
        self.target_type = module_list[-1].target_type
        self.args = Namespace()

    def build_states(self) -> List[AgentStates]:
        return [m.build_states() for m in self.module_list]

    def reset(self) -> None:
        for m in self.module_list:
            m.reset()

    def push(self, segment: Segment, states: Optional[List[Optional[AgentStates]]] = None) -> None:
        states = states if states is not None else [None] * len(self.module_list)
        assert len(states) == len(self.module_list)
        for i, module in enumerate(self.module_list[:-1]):
            segment = module.pushpop(segment, states[i])
        self.module_list[-1].push(segment, states[-1])

    def pop(self, states: Optional[List[Optional[AgentStates]]] = None) -> Segment:
        last = None if states is None else states[-1]
        return self.module_list[-1].pop(last)

    def pushpop(self, segment: Segment, states: Optional[List[Optional[AgentStates]]] = None) -> Segment:
        self.push(segment, states)
        return self.pop(states)
