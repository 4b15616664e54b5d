#!/usr/bin/env node
// commandline.js — the CLI contract of the reference (/root/reference/commandline.js:2-19,41-59): options
// -threshold -span -pyrScale -pyrLevels -winSize -pyrIterations -polyN -polySigma -flags, two positional
// paths, pretty JSON per result on stdout.  The reference's own script cannot run (it requires
// './opticalflow.js', which does not exist, commandline.js:62); this one drives index.js.  No stdout dup2
// trick is needed: the native layer here does not print.
// expect_path / target_path may be two directories (every file of target is paired with the same relative
// path under expect) or two image files.
'use strict';
var FS = require('fs');
var Path = require('path');

var usage =
  '[USAGE]\n' +
  '  node commandline.js [options] expect_path target_path\n' +
  '\n' +
  'ex)\n' +
  '  $ node commandline.js -threshold 1.0 -span 10 test/images test/images2\n';

var supportedOptions = ['threshold', 'span', 'pyrScale', 'pyrLevels', 'winSize', 'pyrIterations', 'polyN',
  'polySigma', 'flags', 'numThreads'];

function getArgs(argv) {
  var positional = [], options = {};
  for (var i = 0; i < argv.length; i++) {
    var m = /^--?([A-Za-z]+)(?:=(.*))?$/.exec(argv[i]);
    if (m && supportedOptions.indexOf(m[1]) !== -1) {
      var v = m[2] !== undefined ? m[2] : argv[++i];
      options[m[1]] = Number(v);          // numbers, like minimist does for numeric-looking values
    } else {
      positional.push(argv[i]);
    }
  }
  if (positional.length < 2) { process.stderr.write(usage); process.exit(-1); }
  return { expect_path: positional[0], target_path: positional[1], options: options };
}

function print(o) { process.stdout.write(JSON.stringify(o, null, '  ') + '\n'); }

function main(args) {
  var TW = require('./index');
  var isDir = false;
  try { isDir = FS.statSync(args.target_path).isDirectory(); } catch (e) { /* reported as an ERROR response */ }
  var t;
  if (isDir) {
    var opts = {};
    Object.keys(args.options).forEach(function(k) { opts[k] = args.options[k]; });
    opts.expectDir = args.expect_path;
    t = TW.create(args.target_path, opts);
  } else {
    t = new TW.TidalWave(args.options);
    t.on('data', function() { t.dispose(); });
    t.on('error', function() { t.dispose(); });
    t.calc(Path.resolve(args.expect_path), Path.resolve(args.target_path));
  }
  t.on('data', print);
  t.on('error', print);
  t.once('finish', print);
}

main(getArgs(process.argv.slice(2)));
