"""Rooted tree with named nodes and branch lengths: the interface of the reference's
`cherryml.io.Tree` (io/_tree.py:7-140) that the counting and SiteRM stages use, plus its text
format (`read_tree` / `write_tree`, io/_tree.py:188-260):

    "<n> nodes\\n" + n lines "<name>\\n" + "<m> edges\\n" + m lines "<parent> <child> <length>\\n"

Children keep insertion order (the pairing rules of cherry++ depend on it)."""
from typing import Dict, List, Tuple


class Tree:
    def __init__(self) -> None:
        self._children: Dict[str, List[Tuple[str, float]]] = {}
        self._parent: Dict[str, Tuple[str, float]] = {}
        self._edges: List[Tuple[str, str, float]] = []

    # -- construction -----------------------------------------------------------------
    def add_node(self, v: str) -> None:
        self._children[v] = []

    def add_nodes(self, nodes: List[str]) -> None:
        for v in nodes:
            self.add_node(v)

    def add_edge(self, u: str, v: str, length: float) -> None:
        if v in self._parent:
            raise Exception(f"Node {v} already has a parent ({self._parent[v][0]}), cannot also have "
                            f"parent {u} - graph is not a tree.")
        self._children[u].append((v, length))
        self._parent[v] = (u, length)
        self._edges.append((u, v, length))

    def add_edges(self, edges: List[Tuple[str, str, float]]) -> None:
        for u, v, length in edges:
            self.add_edge(u, v, length)

    # -- queries ----------------------------------------------------------------------
    def edges(self) -> List[Tuple[str, str, float]]:
        return list(self._edges)

    def nodes(self) -> List[str]:
        return list(self._children)

    def is_node(self, v: str) -> bool:
        return v in self._children

    def root(self) -> str:
        roots = [u for u in self._children if u not in self._parent]
        if len(roots) != 1:
            raise Exception(f"Tree should have one root, but found: {roots}")
        return roots[0]

    def children(self, u: str) -> List[Tuple[str, float]]:
        return list(self._children[u])

    def parent(self, u: str) -> Tuple[str, float]:
        return self._parent[u]

    def is_leaf(self, u: str) -> bool:
        return not self._children[u]

    def is_root(self, u: str) -> bool:
        return u not in self._parent

    def leaves(self) -> List[str]:
        return [u for u in self._children if not self._children[u]]

    def internal_nodes(self) -> List[str]:
        return [u for u in self._children if self._children[u]]

    def num_nodes(self) -> int:
        return len(self._children)

    def num_edges(self) -> int:
        return len(self._edges)

    def _order(self, post: bool) -> List[str]:
        out, stack = [], [(self.root(), False)]
        while stack:
            v, done = stack.pop()
            if done:
                out.append(v)
                continue
            if post:
                stack.append((v, True))
            else:
                out.append(v)
            for c, _ in reversed(self._children[v]):
                stack.append((c, False))
        return out

    def preorder_traversal(self) -> List[str]:
        return self._order(False)

    def postorder_traversal(self) -> List[str]:
        return self._order(True)

    def __str__(self) -> str:
        res = f"Tree with {self.num_nodes()} nodes, and {self.num_edges()} edges:\n"
        for u, ch in self._children.items():
            for v, length in ch:
                res += f"{u} -> {v}: {length}\n"
        return res


def read_tree(tree_path: str) -> Tree:
    with open(tree_path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    tree = Tree()
    n = int(lines[0].split()[0])
    for i in range(n):
        tree.add_node(lines[1 + i].strip())
    try:
        m = int(lines[1 + n].split()[0])
    except Exception:
        raise Exception(f"Tree file: {tree_path} should have an '<m> edges' line after the nodes")
    for i in range(m):
        u, v, length = lines[2 + n + i].split()
        if not tree.is_node(u) or not tree.is_node(v):
            raise Exception(f"In Tree file {tree_path}: {u} and {v} should be nodes in the tree, but aren't.")
        tree.add_edge(u, v, float(length))
    return tree


def write_tree(tree: Tree, tree_path: str, scaling_factor: float = 1.0, node_name_prefix: str = "") -> None:
    out = [f"{tree.num_nodes()} nodes\n"]
    out += [f"{node_name_prefix}{v}\n" for v in tree.nodes()]
    out.append(f"{tree.num_edges()} edges\n")
    out += [f"{node_name_prefix}{u} {node_name_prefix}{v} {length * scaling_factor}\n" for u, v, length in tree.edges()]
    with open(tree_path, "w") as f:
        f.write("".join(out))


def convert_newick_to_CherryML_Tree(tree_newick: str) -> Tree:
    """Newick string -> Tree, as the reference's helper of the same name (io/_tree.py:313-320, which goes
    through ete3 with its default format 0): labels after a closing parenthesis are support values, not
    names, so every internal node is named `internal-<k>` in pre-order (the root is `internal-1`);
    a missing branch length is 1.0; children keep the order of the string (cherry++ pairing depends
    on it).  Own parser: ete3 is not a dependency here."""
    s = tree_newick.strip()
    pos = 0

    def skip_ws():
        nonlocal pos
        while pos < len(s) and (s[pos].isspace() or s[pos] == "["):
            if s[pos] == "[":
                end = s.find("]", pos)
                if end < 0:
                    raise ValueError("newick: unterminated comment")
                pos = end + 1
            else:
                pos += 1

    def label() -> str:
        nonlocal pos
        skip_ws()
        if pos < len(s) and s[pos] in "'\"":
            q = s[pos]
            end = s.find(q, pos + 1)
            if end < 0:
                raise ValueError("newick: unterminated quoted label")
            out = s[pos + 1:end]
            pos = end + 1
            return out
        start = pos
        while pos < len(s) and s[pos] not in "():,;[" and not s[pos].isspace():
            pos += 1
        return s[start:pos]

    def node():
        """returns (name, length, children)"""
        nonlocal pos
        skip_ws()
        children = []
        if pos < len(s) and s[pos] == "(":
            pos += 1
            while True:
                children.append(node())
                skip_ws()
                if pos >= len(s):
                    raise ValueError("newick: unbalanced parentheses")
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"newick: unexpected {s[pos]!r} at {pos}")
        name = label()
        if children:
            name = ""              # ete3 format 0: an internal label is a support value
        length = None
        skip_ws()
        if pos < len(s) and s[pos] == ":":
            pos += 1
            skip_ws()
            length = float(label())
        return name, length, children

    root = node()
    skip_ws()
    if pos < len(s) and s[pos] == ";":
        pos += 1
    skip_ws()
    if pos != len(s):
        raise ValueError(f"newick: trailing characters at {pos}")
    tree = Tree()
    counter = [0]
    # pre-order, iteratively: name, add node, add edge from the parent, then the children in order
    stack = [(None, root)]
    while stack:
        parent, (name, length, children) = stack.pop()
        if name == "":
            counter[0] += 1
            name = f"internal-{counter[0]}"
        tree.add_node(name)
        if parent is not None:
            tree.add_edge(parent, name, 1.0 if length is None else length)
        for child in reversed(children):
            stack.append((name, child))
    return tree
