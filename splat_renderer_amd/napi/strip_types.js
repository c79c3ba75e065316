'use strict';
/**
 * index.ts -> index.js: erases the TypeScript-only syntax index.ts uses (this image has no TypeScript compiler; with one,
 * `tsc --target es2019 --module commonjs` on index.ts is the same step).  index.ts keeps to syntax that erases without
 * moving a character of what remains:
 *   - whole lines:   `declare ...;`  (class fields, ambient require/module),  `type X = ...;`,  `interface X { ... }`
 *   - in the header of a function or of a class member (a line `  name(...)... {` / `function name(...)... {`):
 *     `<T ...>` after the name, `: Type` and `?` after a parameter, `: Type` after the parameter list
 * and nothing else (no `as`, no annotated locals, no enums, no parameter properties).  Usage:
 *   node strip_types.js index.ts > index.js        (tests/test_napi.py checks that the committed twin is exactly this)
 */
const fs = require('fs');

function closer(text, open, i) { // index of the bracket closing the one at i
  const pairs = { '(': ')', '[': ']', '{': '}', '<': '>' };
  const close = pairs[open];
  let depth = 0;
  for (let j = i; j < text.length; j++) {
    const ch = text[j];
    if (ch === open) depth++;
    else if (ch === close && !(ch === '>' && text[j - 1] === '=')) {
      depth--;
      if (depth === 0) return j;
    }
  }
  throw new Error('unbalanced ' + open + ' in: ' + text);
}

function splitTop(text) { // at commas outside every bracket
  const out = [];
  let depth = 0, cur = '';
  for (let j = 0; j < text.length; j++) {
    const ch = text[j];
    if ('([{<'.includes(ch)) depth++;
    else if (')]}'.includes(ch) || (ch === '>' && text[j - 1] !== '=')) depth--;
    if (ch === ',' && depth === 0) {
      out.push(cur);
      cur = '';
    } else cur += ch;
  }
  if (cur.trim() !== '') out.push(cur);
  return out;
}

function eraseParam(param) { // "name?: Type = dflt" -> "name = dflt"
  const m = /^(\s*)(\.\.\.)?([A-Za-z_$][\w$]*)(\?)?\s*:/.exec(param);
  if (!m) return param; // untyped (or destructured: index.ts leaves those untyped)
  const head = m[1] + (m[2] || '') + m[3];
  let depth = 0;
  for (let j = m[0].length; j < param.length; j++) { // the type ends at a `=` outside brackets that is not `=>`
    const ch = param[j];
    if ('([{<'.includes(ch)) depth++;
    else if (')]}'.includes(ch) || (ch === '>' && param[j - 1] !== '=')) depth--;
    if (ch === '=' && depth === 0 && param[j + 1] !== '>') return head + ' ' + param.slice(j);
  }
  return head;
}

function eraseHeader(line, nameEnd) { // nameEnd: index just after the member / function name
  let i = nameEnd;
  let generic = '';
  if (line[i] === '<') {
    const g = closer(line, '<', i);
    generic = line.slice(i, g + 1);
    i = g + 1;
  }
  if (line[i] !== '(') return line;
  const e = closer(line, '(', i);
  let rest = line.slice(e + 1);
  if (!rest.startsWith(' {') && !rest.startsWith(':')) return line; // a call, not a header
  const params = splitTop(line.slice(i + 1, e)).map(eraseParam).join(',');
  if (rest.startsWith(':')) { // return type: up to the ` {` that opens the body (index.ts uses no object-literal return types)
    const b = rest.indexOf(' {');
    if (b < 0) throw new Error('no body after the return type: ' + line);
    if (rest.slice(1, b).includes('{')) throw new Error('object-literal return type: ' + line);
    rest = rest.slice(b);
  }
  void generic;
  return line.slice(0, nameEnd) + '(' + params + ')' + rest;
}

function strip(source) {
  const out = [];
  const lines = source.split('\n');
  for (let n = 0; n < lines.length; n++) {
    const line = lines[n];
    const t = line.trim();
    if (t.startsWith('declare ')) continue;
    if (/^interface [\w$]+/.test(line) || /^type [\w$]+( |<)/.test(line)) { // to the end of the statement
      let depth = 0, k = n;
      for (;; k++) {
        for (const ch of lines[k]) {
          if (ch === '{') depth++;
          else if (ch === '}') depth--;
        }
        if (depth === 0 && (lines[k].trimEnd().endsWith(';') || lines[k].trimEnd().endsWith('}'))) break;
      }
      n = k;
      continue;
    }
    let m = /^function ([A-Za-z_$][\w$]*)/.exec(line);
    if (m) {
      out.push(eraseHeader(line, m[0].length));
      continue;
    }
    m = /^ {2}((?:static |get |set |async )?)([A-Za-z_$][\w$]*)(?=[(<])/.exec(line);
    if (m && !['if', 'for', 'while', 'switch', 'return', 'super', 'catch'].includes(m[2])) {
      out.push(eraseHeader(line, m[0].length));
      continue;
    }
    out.push(line);
  }
  return out.join('\n');
}

if (require.main === module) process.stdout.write(strip(fs.readFileSync(process.argv[2], 'utf8')));
module.exports = { strip };
